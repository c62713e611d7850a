import cProfile, pstats, json, os, sys, tempfile, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
from nano_qwen3tts_vllm.sampling_params import SamplingParams
from nano_qwen3tts_vllm.utils import synthetic
dev = torch.device("cuda", 0)
cfg = synthetic.make_config("1.7B")
tw, pw = bench.make_weights(cfg, dev)
d = tempfile.mkdtemp()
json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
fe = FrameEngine(d, max_streams=2, max_text_frames=16, talker_weights=tw, predictor_weights=pw, num_kvcache_blocks=8, max_num_batched_tokens=4096, codec_eos=None)
sp = SamplingParams(temperature=1.0, max_tokens=1, do_sample=False)
sp2 = SamplingParams(temperature=0.9, max_tokens=17, do_sample=False)
plen = int(sys.argv[1])
def once():
    x = torch.randn(plen, 2048, device=dev) * 0.5
    t = torch.randn(8, 2048, device=dev) * 0.1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sid = fe.add_stream(x, t, 8, sp, sp2)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    fe.remove_stream(sid)
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3
for _ in range(3): once()
r = [once() for _ in range(10)]
print("host-return ms", [round(a, 2) for a, b in r])
print("synced ms     ", [round(b, 2) for a, b in r])
pr = cProfile.Profile()
pr.enable()
for _ in range(20): once()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
