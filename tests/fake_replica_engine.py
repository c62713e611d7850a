"""A deterministic stand-in for FrameEngine (same protocol: add_stream / step / remove_stream / streams) that the
replica-pool tests run in CPU processes.  A stream's frame f is 16 ids derived from its prompt's checksum, the frame
index and the replica's tag, so a test can tell which replica served a stream and that frames arrive in order."""
import os
import time


class NoCapacity(RuntimeError):
    pass


class _S:
    def __init__(self, key, max_frames):
        self.key, self.max_frames, self.n = key, max_frames, 0


class FakeEngine:
    def __init__(self, tag=0, max_streams=2, step_seconds=0.002, fail_on=None, die_on=None, paced=False):
        self.tag, self.max_streams, self.step_seconds, self.fail_on = tag, max_streams, step_seconds, fail_on
        self.die_on = die_on
        # paced: frames on a fixed schedule (frame k ends at start + k * step_seconds; a late wake-up shortens the next
        # wait), so that the offered rate of a load test is exactly streams / step_seconds
        self.paced, self._deadline = paced, None
        self.streams = {}
        self.visible = os.environ.get("HIP_VISIBLE_DEVICES")

    def add_stream(self, prompt_embeds, text_adds, max_frames, talker_sampling=None, predictor_sampling=None, sid=None):
        if len(self.streams) >= self.max_streams:
            raise NoCapacity("all stream rows are taken")
        if prompt_embeds.shape[0] + max_frames > 1024:
            raise AssertionError("prompt + frames exceed max_model_len (1024)")
        self.streams[sid] = _S(int(prompt_embeds.float().sum().item()) % 1000, max_frames)
        return sid

    def remove_stream(self, sid):
        self.streams.pop(sid, None)

    def step(self):
        if self.paced:
            now = time.perf_counter()
            if self._deadline is None or now - self._deadline > 10 * self.step_seconds:      # first frame, or after idling
                self._deadline = now
            self._deadline += self.step_seconds
            if self._deadline > now:
                time.sleep(self._deadline - now)
        else:
            time.sleep(self.step_seconds)
        out = {}
        for sid, s in list(self.streams.items()):
            if self.fail_on is not None and s.key == self.fail_on:
                raise RuntimeError("injected engine failure")
            if self.die_on is not None and s.key == self.die_on:
                os._exit(17)                               # the whole replica process goes away mid-stream
            dev = int(self.visible) if self.visible not in (None, "") else -1
            out[sid] = [s.key, s.n, dev, self.tag] + [0] * 12
            s.n += 1
            if s.n >= s.max_frames:
                del self.streams[sid]
        return out

    def exit(self):
        pass


def make(**kw):
    return FakeEngine(**kw)


def broken(**kw):
    raise RuntimeError("no engine today")


class FakePollingEngine(FakeEngine):
    """With the real engine's serving hooks: add_streams(wait=False) and step(poll=...), which calls the hook while its
    frame is 'in flight'.  Frames carry, in slot 4, whether the stream was admitted from inside a step."""
    overlapped_admission = True
    join_prefills = True

    def __init__(self, **kw):
        super().__init__(**kw)
        self._in_step = False

    def add_streams(self, requests, wait=True):
        out = []
        for r in requests:
            try:
                sid = self.add_stream(**r)
                self.streams[sid].in_flight = self._in_step
                out.append(sid)
            except Exception as e:
                out.append(e)
        return out

    def step(self, poll=None):
        if poll is not None:
            self._in_step = True
            poll()
        out = super().step()
        for sid, fr in out.items():
            fr[4] = int(getattr(self.streams.get(sid), "in_flight", False)) if sid in self.streams else fr[4]
        if poll is not None:
            for _ in range(3):
                time.sleep(self.step_seconds / 3)
                poll()
            self._in_step = False
        return out


def make_polling(**kw):
    return FakePollingEngine(**kw)
