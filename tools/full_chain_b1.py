"""Side experiment: B=1 tokens -> waveform, for profiling the phoneme-rate path's launch count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylish_tts_amd import params, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.pipeline import Synthesizer
from stylish_tts_amd.runtime import HipModel
cfg = load_model_config()
w = {m: params.synth_state_dict(params.module_spec(m, cfg), 0, prefix=m + ".") for m in params.MODULE_SPECS}
eng = HipModel(cfg, 0); eng.load_weights(w, which=255)
syn = Synthesizer(eng, adapt=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
toks = [synth.tokens(f"fc.{B}.{i}", 1, 14, 178)[0].tolist() for i in range(B)]
for _ in range(3): syn(toks)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
for _ in range(n): syn(toks)
torch.cuda.synchronize(); print("ms/call", (time.perf_counter() - t0) / n * 1e3)
