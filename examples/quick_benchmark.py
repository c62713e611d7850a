"""One request at a time through the API object: the counterpart of the reference's examples/quick_benchmark.py
(E/quick_benchmark.py:25-84: from_pretrained -> warm-up -> N timed generate_voice_design(text, instruct) calls ->
audio seconds from the frame count -> average time and RTF).  Text goes in as there; on an offline box the
released tokenizer is replaced by a character-level stand-in and the weights are random, so the audio is
noise and generation runs to a fixed frame count instead of the codec's end token.  The codec decode
(codes -> PCM) is not part of this build and is not timed.

    python examples/quick_benchmark.py [--size 1.7B] [--runs 3] [--frames 125] [--sample] [--embeds]
"""
import argparse
import json
import statistics
import time

from _common import FRAME_SECONDS, CharProcessor, model_dir, request

TEXT = "The quick brown fox jumps over the lazy dog while the evening train rolls slowly out of the station."
INSTRUCT = "A warm, unhurried male voice in his forties, speaking clearly."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1.7B")
    ap.add_argument("--model-dir", default=None)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--frames", type=int, default=125, help="frames per request (10 s of audio); codec EOS is ignored")
    ap.add_argument("--prompt-len", type=int, default=48)
    ap.add_argument("--sample", action="store_true", help="the reference's sampling (T=1.0/0.9, top-k 50) instead of greedy")
    ap.add_argument("--embeds", action="store_true", help="skip the text front-end: hand synthetic embeddings to generate()")
    args = ap.parse_args()

    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    d = model_dir(args.size, args.model_dir)
    iface = Qwen3TTSInterface.from_pretrained(d, enforce_eager=False, tensor_parallel_size=1)
    iface.do_sample = args.sample
    iface.ignore_eos = True
    H = iface.model_config.talker_config.hidden_size
    prompt, trailing, pad = request(H, args.prompt_len, 20, seed=1000)
    if iface.processor is None:
        iface.processor = CharProcessor(iface.model_config.talker_config.text_vocab_size)

    def start(n):
        if args.embeds:
            return iface.generate(prompt, trailing, pad, max_frames=n)
        iface.max_frames = n
        return iface.generate_voice_design(text=TEXT, instruct=INSTRUCT, language="English")

    list(start(8))                                                            # warm-up: graphs captured here
    times, first = [], []
    n_frames = 0
    for _ in range(args.runs):
        t0 = time.perf_counter()
        gen = start(args.frames)
        frames = [next(gen)]
        first.append(time.perf_counter() - t0)
        frames += list(gen)
        times.append(time.perf_counter() - t0)
        n_frames = len(frames)
        assert all(len(f) == 16 for f in frames)
    audio_s = n_frames * FRAME_SECONDS
    avg = statistics.mean(times)
    print(json.dumps({"example": "quick_benchmark", "model": args.size, "runs": args.runs, "frames": n_frames,
                      "audio_seconds": audio_s, "avg_generation_s": avg, "rtf": avg / audio_s,
                      "first_frame_ms_p50": statistics.median(first) * 1e3,
                      "codec_tokens_per_s": 16 * n_frames / avg, "sampling": "reference" if args.sample else "greedy",
                      "entry": "generate(embeds)" if args.embeds else "generate_voice_design(text, instruct)",
                      "data": "random-init weights, character-level stand-in tokenizer; codec decode not included"}))
    iface.shutdown()


if __name__ == "__main__":
    main()
