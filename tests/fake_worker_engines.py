"""Deterministic stand-ins for TalkerLLM / PredictorLLM with the surface the worker command loops use
(add_request / step_with_outputs / step / clear_request / scheduler.waiting|running / exit).  They live in a module of
their own so that a spawned worker process can import them by name (workers.*_worker.run_*_worker(engine_factory=...))."""
import torch


class FakeTalker:
    """Token = hash of the fed embedding; hidden = mean row.  Same API as TalkerLLM."""

    def __init__(self, *a, **k):
        self.seqs, self.cleared, self.exited = {}, [], False

    def add_request(self, inputs_embeds, sampling_params, request_id=None):
        for e in inputs_embeds:
            st = self.seqs.setdefault(request_id, {"toks": [], "pending": None, "sid": len(self.seqs)})
            st["pending"] = e.float().reshape(-1, e.shape[-1])

    def clear_request(self, request_id):
        self.cleared.append(request_id)
        self.seqs.pop(request_id, None)

    def step_with_outputs(self):
        out = []
        for rid, st in self.seqs.items():
            if st["pending"] is None:
                continue
            x = st["pending"]
            st["pending"] = None
            tok = int(x.abs().sum().item() * 1000) % 500
            st["toks"].append(tok)
            out.append((rid, st["sid"], list(st["toks"]), x.mean(0).to(torch.bfloat16), False))
        return [], -len(out), out

    def exit(self):
        self.exited = True


class _Sched:
    def __init__(self):
        self.waiting, self.running = [], []


class FakePredictor:
    def __init__(self, *a, **k):
        self.scheduler, self.cleared, self.exited = _Sched(), [], False

    def add_request(self, inputs_embeds, sampling_params, request_id=None):
        for e in inputs_embeds:
            self.scheduler.waiting.append((request_id, e.float()))

    def clear_request(self, request_id):
        self.cleared.append(request_id)

    def step(self):
        outs = []
        for i, (rid, e) in enumerate(self.scheduler.waiting):
            base = int(e.abs().sum().item() * 100)
            outs.append((rid, i, [(base + 7 * k) % 256 for k in range(15)]))
        self.scheduler.waiting.clear()
        return outs, len(outs)

    def exit(self):
        self.exited = True
