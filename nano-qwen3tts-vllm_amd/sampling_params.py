"""Per-request sampling options (reference: P/sampling_params.py:4-14).

`do_sample`, `top_k` and `top_p` travel with every request in the reference but
its sampler ignores them (P/layers/sampler.py:15); here `do_sample=False` selects
the greedy path (argmax of the fp32 logits, lowest index on ties) and `top_k` is
honoured.  The temperature floor of the reference is kept."""
from dataclasses import dataclass


@dataclass
class SamplingParams:
    temperature: float = 1.0
    max_tokens: int = 64
    ignore_eos: bool = False
    do_sample: bool = True
    top_k: int = 50
    top_p: float = 1.0

    def __post_init__(self):
        assert self.temperature > 1e-10, "greedy sampling is not permitted"
