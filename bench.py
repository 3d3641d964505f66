#!/usr/bin/env python3
"""bench.py — utterances/s and RTF of the frame-rate hot path on MI355X (BASELINE.json cfg2).

    python bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (`stts_frame_path`: Decoder → PriorEncoder + reverse flow + post_flow →
harmonic source → STFT → vocoder body → iSTFT → tanh) over one batch of 8 synthetic LJSpeech-shaped utterances
of 3.0 s (T = 240 mel frames, T4 = 960 vocoder frames, 72 000 samples) in fp32, inputs resident in HBM.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); rank 0 broadcasts the packed weights once,
every rank runs its own batch (weak scaling, utterances are independent) and the waveforms are gathered to rank 0
inside the timed step.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N = 1): `roofline` for the dominant kernels (conv_gemm_f32 + wn_layer_kernel) from HIP start/stop events on
every launch on its own stream in a profiling pass right after the timed region, and `cpu_baseline` = the numpy oracle
timed on the host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

BATCH, T_MEL, SR = 8, 240, 24000
T4 = 4 * T_MEL
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_PEAK_TFLOPS = {"f32": FP32_MFMA_PEAK_TFLOPS, "bf16": 2500.0, "f16": 2500.0}  # dense 16-bit MFMA (same guide)


def synth_inputs(rank: int, device):
    from stylish_tts_amd import synth

    tag = f"bench.r{rank}"
    asr = np.concatenate([synth.normal(f"{tag}.asr{b}", (T4, 128)) for b in range(BATCH)])
    pitch = np.concatenate([synth.pitch_curve(f"{tag}.pitch{b}", 1, T4)[0] for b in range(BATCH)])
    energy = np.concatenate([(synth.uniform(f"{tag}.energy{b}", (T4,)) * 2 + 2).astype(np.float32) for b in range(BATCH)])
    style = (synth.normal(f"{tag}.style", (BATCH, 64)) * 0.7).astype(np.float32)
    nz = synth.path_noise(tag, BATCH, T4)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)  # noqa: E731
    return dict(
        asr=d(asr), pitch=d(pitch), energy=d(energy), style=d(style),
        prior_noise=d(nz["prior_noise"].transpose(0, 2, 1).reshape(BATCH * T4, 128)),
        src_noise=d(nz["src_noise"].reshape(-1)), init_phase=d(nz["init_phase"].reshape(-1)),
        host=dict(asr=asr, pitch=pitch, energy=energy, style=style, nz=nz),
    )


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU per step (BASELINE cfg2 = 8; other values are side experiments)")
    ap.add_argument("--mel-frames", type=int, default=240, help="mel frames per utterance (cfg2 = 240 = 3.0 s)")
    ap.add_argument("--precision", choices=["f32", "bf16", "f16"], default="f32",
                    help="operand precision of the contractions; f32 = BASELINE cfg2 (the bench line), bf16 / f16 = cfg3 / cfg5 arithmetic (side experiments)")
    ap.add_argument("--cpu-utts", type=int, default=4, help="utterances the CPU baseline times (bounded sample)")
    args = ap.parse_args()
    global BATCH, T_MEL, T4
    BATCH, T_MEL = args.batch, args.mel_frames
    T4 = 4 * T_MEL

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    import torch.distributed as dist

    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import __graft_entry__ as entry

    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()

    from stylish_tts_amd import _lib, params
    from stylish_tts_amd.config import load_model_config
    from stylish_tts_amd.runtime import HipModel, Segments
    from stylish_tts_amd.sharding import broadcast_state_dict

    cfg = load_model_config()
    spec = params.module_spec("speech_predictor", cfg)
    sd = params.synth_state_dict(spec, 0, prefix="speech_predictor.") if rank == 0 else None
    if world > 1:
        sd = broadcast_state_dict(sd, spec, device, src=0)  # RCCL broadcast of the weights, once
    model = HipModel(cfg, local, precision=args.precision)
    model.load_weights({"speech_predictor": sd}, which=7)

    seg = Segments([T4] * BATCH, device)
    inp = synth_inputs(rank, device)
    audio = torch.empty(BATCH * T4 * 75, dtype=torch.float32, device=device)
    gathered = [torch.empty_like(audio) for _ in range(world)] if (world > 1 and rank == 0) else None

    collect = {"mode": "gather"}
    all_buf = torch.empty(world * audio.numel(), dtype=torch.float32, device=device) if world > 1 else None

    def step():
        model.frame_path(seg, inp["asr"], inp["pitch"], inp["energy"], inp["style"], inp["prior_noise"], inp["src_noise"], inp["init_phase"],
                         batch_scope=True, out=audio)
        if world > 1:
            if collect["mode"] == "gather":
                dist.gather(audio, gathered, dst=0)  # waveforms to rank 0 over xGMI
            else:
                dist.all_gather_into_tensor(all_buf, audio)

    if world > 1:
        # pick the collective once, outside the timed region: gather (rank 0 receives) unless this RCCL build lacks it
        ok = torch.ones(1, device=device)
        try:
            dist.gather(audio, gathered, dst=0)
            torch.cuda.synchronize()
        except Exception as e:  # pragma: no cover
            ok.zero_()
            print(f"[bench] rank {rank}: dist.gather unavailable ({e}); using all_gather_into_tensor", file=sys.stderr)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() < 1:
            collect["mode"] = "all_gather"

    for _ in range(args.warmup):
        step()
    model.check_status()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(audio).all()

    utts = world * BATCH * args.steps
    value = utts / elapsed
    audio_seconds = utts * (T4 * 75 / SR)
    out = {
        "metric": "utterances_per_sec",
        "value": round(value, 3),
        "unit": "utt/s",
        "rtf": elapsed / audio_seconds,
        "realtime_x": audio_seconds / elapsed,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {
            "workload": ("cfg2: " if (BATCH, T_MEL, args.precision) == (8, 240, "f32") else "side experiment: ") + f"LJSpeech-shaped batch={BATCH} x {T_MEL / 80:.1f} s (T={T_MEL} mel frames, {T4 * 75} samples @24 kHz) " + ("fp32" if args.precision == "f32" else f"{args.precision} matrix-core operands, fp32 accumulate") + ", "
                        "Decoder + PriorEncoder/reverse flow + freegan iSTFT vocoder (stts_frame_path); no diffusion step exists in the reference",
            "batch_per_gpu": BATCH,
            "global_batch": BATCH * world,
            "frames_per_utt": T4,
            "parallelism": f"utterance-sharded x{world}; RCCL broadcast(weights, once) + gather(waveforms, per step)" if world > 1 else "single GPU",
        },
    }

    if rank == 0 and world == 1:
        # ---- roofline leg: HIP events around every conv_gemm_f32 launch (same stream), 3 steps right after the timed region
        lib = _lib.load()
        psteps = 3
        lib.stts_profile_begin()
        for _ in range(psteps):
            step()
        n, ms, fl = C.c_int(), C.c_double(), C.c_double()
        _lib.check(lib.stts_profile_end(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(n), C.byref(ms), C.byref(fl)))
        launches = n.value // psteps
        avg_ms = ms.value / max(n.value, 1)
        achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {
            "kernel": "conv_gemm_f32 + wn_layer_kernel (all Conv1d/Linear contractions of the step; Winograd-form convs timed with their transforms)",
            "bound": "mfma",
            "achieved": round(achieved, 2),
            "peak": MFMA_PEAK_TFLOPS[args.precision],
            "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS[args.precision], 4),
            "traffic": traffic,
            "launches_per_step": launches,
            "avg_launch_ms": round(avg_ms, 5),
            "algorithmic_gflop_per_step": round(fl.value / psteps / 1e9, 2),
            "gemm_ms_per_step": round(ms.value / psteps, 4),
            "gemm_share_of_step": round((ms.value / psteps) / (1e3 * elapsed / args.steps), 4),
        }
        # ---- CPU baseline leg: the oracle (validated against reference goldens) on the host cores
        if not args.no_cpu_baseline:
            from oracle import stylish_oracle as O

            h = inp["host"]
            n_utts = max(1, args.cpu_utts)
            t1 = time.perf_counter()
            for b in range(n_utts):
                bb = b % BATCH
                sl = slice(bb * T4, (bb + 1) * T4)
                nzb = dict(prior_noise=h["nz"]["prior_noise"][bb : bb + 1], src_noise=h["nz"]["src_noise"][bb : bb + 1], init_phase=h["nz"]["init_phase"])
                O.frame_path(h["asr"][sl].T[None].copy(), h["pitch"][sl][None], h["energy"][sl][None], h["style"][bb : bb + 1], nzb, sd)
            cpu_t = time.perf_counter() - t1
            try:  # the threads the oracle's matrix products actually ran on
                from threadpoolctl import threadpool_info

                blas_threads = max([int(p.get("num_threads", 1)) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
            except Exception:
                blas_threads = os.cpu_count()
            out["cpu_baseline"] = {
                "value": round(n_utts / cpu_t, 4),
                "unit": "utt/s",
                "cores": blas_threads,
                "kind": "port",
                "sample": f"{n_utts} utterances of the same workload (3.0 s each, B=1 per call), numpy oracle; BLAS pool = {blas_threads} threads "
                          f"of {os.cpu_count()} logical cores",
                "reference_torch_cpu_note": "survey container, 8 vCPU, reference torch-CPU code: 1.4 utt/s at B=1, 1.9-2.5 utt/s at B=8 (BASELINE.md §2)",
            }
    if rank == 0:
        print(json.dumps(out), flush=True)
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
