import os, sys
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: F401  (the -DQTTS_TUNING build of the library)
from nano_qwen3tts_vllm import hip
hip.lib.qtts_debug_linear(int(os.environ.get("QTTS_LINEAR_DBG", "0")))
if os.environ.get("QTTS_SUB_MIN_BYTES"):
    hip.lib.qtts_tune_subtiles(int(os.environ["QTTS_SUB_MIN_BYTES"]))   # smallest weight slice of a workgroup that shares a tile
if os.environ.get("QTTS_TUNE_T"):
    hip.lib.qtts_tune_linear(0, int(os.environ["QTTS_TUNE_T"]))         # decode GEMM: workgroups per CU x 64 as the tile-loop target
_tuning.apply_env_flags()
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
bench.main()
