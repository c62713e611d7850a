"""Public engine classes (reference: P/llm.py:1-9)."""
from .engine.llm_engine.predictor_llm_engine import PredictorLLMEngine
from .engine.llm_engine.talker_llm_engine import TalkerLLMEngine


class TalkerLLM(TalkerLLMEngine):
    pass


class PredictorLLM(PredictorLLMEngine):
    pass
