"""Host-side guard for engines that are driven by one thread at a time.

The frame engine orders its HIP streams with events that are always recorded before they are waited for, in the
order the host issues the calls: every cross-stream dependency points backwards in that one order, so the streams
cannot wait for each other in a cycle.  That argument needs ONE host order.  Two threads inside the engine at once
(a serving loop's worker thread stepping while another thread hands a stream back) interleave records and waits on
shared events and staging slots; this guard turns that into an error at the call instead of a hang on the GPU."""
from __future__ import annotations

import threading


class SingleCaller:
    """Re-entrant for the thread that is inside (the engine's `poll` hook admits streams from within `step`), an
    error for any other thread that arrives meanwhile."""

    def __init__(self, what: str):
        self._lock = threading.RLock()
        self._what = what

    def __enter__(self):
        if not self._lock.acquire(blocking=False):
            raise RuntimeError(f"{self._what} is being used by another thread: it is single-threaded by design "
                               f"(one host order of stream records and waits); route every call through one thread")
        return self

    def __exit__(self, *exc):
        self._lock.release()
        return False
