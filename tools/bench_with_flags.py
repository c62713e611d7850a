import os, sys
sys.path.insert(0, "/root/repo")
from nano_qwen3tts_vllm import hip
hip.lib.qtts_debug_linear(int(os.environ.get("QTTS_LINEAR_DBG", "0")))
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
bench.main()
