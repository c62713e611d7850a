"""A CPU stand-in for engine.frame_engine.FrameEngine with the surface Qwen3TTSInterface uses (constructor keywords,
`tr.config.max_model_len`, `max_streams`, `streams`, `capture_all`, `add_stream`, `step`, `remove_stream`, `exit`).
Frames are deterministic functions of the prompt; a stream ends after `FRAMES` frames, as if the talker had emitted
the codec EOS id.  Used where the API surface is exercised without a GPU."""
from types import SimpleNamespace

FRAMES = 12


class NoCapacity(RuntimeError):
    pass


class FakeFrameEngine:
    overlapped_admission = False

    def __init__(self, model_dir, max_streams=8, enforce_eager=False, codec_eos=2150, **kw):
        self.model_dir, self.max_streams, self.enforce_eager, self.codec_eos = model_dir, max_streams, enforce_eager, codec_eos
        self.tr = SimpleNamespace(config=SimpleNamespace(max_model_len=1024))
        self.streams = {}
        self.captured = False
        self.kw = kw

    def capture_all(self, max_context=None):
        self.captured = True
        return 0

    def add_stream(self, prompt_embeds, text_adds, max_frames, talker_sampling=None, predictor_sampling=None, sid=None,
                   wait=True):
        if len(self.streams) >= self.max_streams:
            raise NoCapacity("all stream rows are taken")
        assert prompt_embeds.dim() == 2 and text_adds.dim() == 2 and text_adds.shape[1] == prompt_embeds.shape[1]
        assert prompt_embeds.shape[0] + max_frames + 1 <= 1024, "prompt + frames exceed max_model_len (1024)"
        key = int(prompt_embeds.float().abs().sum().item() * 7) % 1000
        self.streams[sid] = SimpleNamespace(key=key, n=0, limit=min(max_frames, FRAMES))
        return sid

    def add_streams(self, requests):
        out = []
        for r in requests:
            try:
                out.append(self.add_stream(**r))
            except (NoCapacity, AssertionError, ValueError) as e:
                out.append(e)
        return out

    def remove_stream(self, sid):
        self.streams.pop(sid, None)

    def step(self):
        out = {}
        for sid, s in list(self.streams.items()):
            if s.n >= s.limit:                      # "codec EOS": the stream ends without yielding this frame
                del self.streams[sid]
                continue
            out[sid] = [(s.key + s.n * 17 + c) % 256 for c in range(16)]
            s.n += 1
        return out

    def exit(self):
        self.streams.clear()


class FakePollingFrameEngine(FakeFrameEngine):
    """The same with the serving hooks of the real engine: add_streams(wait=False) and step(poll=...), which calls the
    hook while its 'frame is in flight' (a short sleep), so that requests queued meanwhile are admitted from inside
    step.  Streams admitted from the hook before the frame is 'committed' (the first hook call) take part in it."""
    overlapped_admission = True
    join_prefills = True

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.admitted_in_flight = 0
        self._in_step = False

    def add_streams(self, requests, wait=True):
        if self._in_step:
            self.admitted_in_flight += sum(1 for _ in requests)
        return super().add_streams(requests)

    def step(self, poll=None):
        import time
        if poll is not None:
            self._in_step = True
            poll()                                   # (before the frame is committed: newcomers are part of it)
        out = super().step()
        if poll is not None:
            for _ in range(3):                       # the frame is in flight
                time.sleep(0.002)
                poll()
            self._in_step = False
        return out


class FakeLateJoinFrameEngine(FakePollingFrameEngine):
    """The same, with admissions that do not join at once: a stream admitted without waiting stays in `streams` but
    takes no part in the next `LATE` frames (the real engine's `Stream.active == False`: its prefill still runs beside
    the frames, or waits behind another group's prefill)."""
    LATE = 2

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.late_frames = 0

    def add_streams(self, requests, wait=True):
        before = set(self.streams)
        out = super().add_streams(requests, wait=wait)
        if not wait:
            for sid in set(self.streams) - before:
                self.streams[sid].hold = self.LATE
        return out

    def step(self, poll=None):
        held = {sid: s for sid, s in self.streams.items() if getattr(s, "hold", 0) > 0}
        for sid in held:                             # not part of this frame
            del self.streams[sid]
        try:
            out = super().step(poll=poll)
        finally:
            for sid, s in held.items():
                s.hold -= 1
                self.late_frames += 1
                self.streams[sid] = s
        return out
