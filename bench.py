#!/usr/bin/env python3
"""bench.py — utterances/s and RTF of the frame-rate hot path on MI355X (BASELINE.json cfg2).

    python bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (`stts_frame_path`: Decoder → PriorEncoder + reverse flow + post_flow →
harmonic source → STFT → vocoder body → iSTFT → tanh) over one batch of 8 synthetic LJSpeech-shaped utterances
of 3.0 s (T = 240 mel frames, T4 = 960 vocoder frames, 72 000 samples) in fp32, inputs resident in HBM.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); rank 0 broadcasts the packed weights once,
every rank runs its own batch (weak scaling, utterances are independent) and the waveforms are gathered to rank 0
inside the timed step.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N = 1):
  * `roofline`: HIP start/stop events on EVERY launch of the step (its own stream) in a profiling pass right after the timed
    region (`stts_profile_begin/report`): the dense contractions (conv_gemm_f32, the Winograd-form convs, the fused WaveNet
    kernel) against the MFMA roof, in algorithmic (direct-conv) flops AND in the flops the matrix cores execute; the
    bandwidth-bound kernels (`hbm_kernels`) against 8 TB/s with their algorithmic bytes; launches and
    microseconds per launch of the whole step.
  * `roofline.traffic`: HBM-side bytes per contraction launch from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE)
    over this same script, run as child processes BEFORE this process touches the GPU; null when rocprofv3 is missing.
  * `cpu_baseline`: the numpy oracle timed on the host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
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


CONTRACTION_KERNELS = ("conv_gemm_f32", "wn_fused_kernel", "wn_layer", "winograd_")


def measure_traffic(extra_args):
    """HBM-side bytes per contraction launch: two separate `rocprofv3 --pmc` passes over `bench.py --pmc-child` (child
    processes, started before this process initialises the GPU).  Units and the gfx950 correction follow
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KB, and FETCH_SIZE tallies 128-byte
    requests at 64 bytes (x2).  Returns (bytes_per_launch or None, note)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    out = {}
    with tempfile.TemporaryDirectory(prefix="stts_pmc_", dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                   "--steps", "2", "--warmup", "1", *extra_args]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240)
            except Exception as e:  # noqa: BLE001
                return None, f"rocprofv3 --pmc {counter} failed: {e}"
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter}: rc {r.returncode}, {r.stderr.decode('utf-8', 'replace')[-200:]}"
            tot, n = 0.0, 0
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != counter or not any(k in row["Kernel_Name"] for k in CONTRACTION_KERNELS):
                    continue
                tot += float(row["Counter_Value"])
                n += "winograd_" not in row["Kernel_Name"]  # a Winograd-form conv's transforms belong to its contraction launch
            if n == 0:
                return None, f"no contraction kernel in the {counter} pass"
            out[counter] = tot * 1024 * (2 if counter == "FETCH_SIZE" else 1) / n
    return int(out["FETCH_SIZE"] + out["WRITE_SIZE"]), ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this script (--steps 2), per contraction-kernel "
                                                       "dispatch; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); L2-miss traffic incl. Infinity-Cache hits")


WORKLOADS = {
    # BASELINE.json configs[3]: 256 utterances of 0.25-10 s (seeded), fp32, one fixed global batch sharded over the ranks
    "cfg4": dict(n_utt=256, seconds=(0.25, 10.0), precision="f32", note="mixed-length batch: 256 utterances of 0.25-10 s"),
    # BASELINE.json configs[4]: 512 segments of 10 s, fp16 operands (the speaker-id input does not exist in the reference: SURVEY.md 8c)
    "cfg5": dict(n_utt=512, seconds=(10.0, 10.0), precision="f16", note="long-form batch: 512 segments of 10 s"),
}
MAX_ROWS_PER_CALL = 220_000  # frames per stts_frame_path call (~12 GB of workspace; 64 x 10 s = 204 800 is the size the tests pin)


def run_sharded(args, world, rank, local, device, dist):
    """--workload cfg4 | cfg5: ONE fixed global batch, utterances partitioned over the ranks by frame count (longest first),
    every rank runs its shard through stts_frame_path (in calls of at most MAX_ROWS_PER_CALL frames) and the waveforms are
    collected on rank 0 with exact sizes (sharding.WaveformCollector) inside the timed step.  Strong scaling: the work is
    fixed, `value` = utterances of the global batch / step time."""
    import __graft_entry__ as entry

    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    from stylish_tts_amd import params
    from stylish_tts_amd.config import load_model_config
    from stylish_tts_amd.runtime import HipModel, Segments
    from stylish_tts_amd.sharding import WaveformCollector, broadcast_state_dict, partition_utterances

    wl = WORKLOADS[args.workload]
    precision = wl["precision"] if args.precision == "f32" and args.workload == "cfg5" else args.precision
    cfg = load_model_config()
    spec = params.module_spec("speech_predictor", cfg)
    sd = params.synth_state_dict(spec, 0, prefix="speech_predictor.") if rank == 0 else None
    if world > 1:
        sd = broadcast_state_dict(sd, spec, device, src=0)
    model = HipModel(cfg, local, precision=precision)
    model.load_weights({"speech_predictor": sd}, which=7)

    n_utt = args.utterances or wl["n_utt"]
    rng = np.random.default_rng(4)
    secs = rng.uniform(wl["seconds"][0], wl["seconds"][1], n_utt)
    frames = [int(4 * max(1, round(80 * v))) for v in secs]  # vocoder frames per utterance (T4 = 4 T)
    parts = partition_utterances(frames, world)
    mine = parts[rank]
    # this rank's calls: consecutive utterances of its shard, at most MAX_ROWS_PER_CALL frames each
    calls, cur, rows = [], [], 0
    for i in mine:
        if cur and rows + frames[i] > MAX_ROWS_PER_CALL:
            calls.append(cur)
            cur, rows = [], 0
        cur.append(i)
        rows += frames[i]
    if cur:
        calls.append(cur)
    gen = torch.Generator(device=device)

    def rand(shape, seed, normal=True):
        gen.manual_seed(seed)
        return (torch.randn if normal else torch.rand)(shape, generator=gen, device=device, dtype=torch.float32)

    batches = []
    for ci, ids in enumerate(calls):
        L = [frames[i] for i in ids]
        R = sum(L)
        seed = 1000 * rank + ci
        pitch = 80.0 + 220.0 * rand((R,), seed + 1, normal=False)
        pitch = torch.where(rand((R,), seed + 2, normal=False) < 0.3, torch.zeros_like(pitch), pitch)  # ~30 % unvoiced frames
        batches.append(dict(seg=Segments(L, device), asr=rand((R, 128), seed + 3), pitch=pitch, energy=2 + 2 * rand((R,), seed + 4, normal=False),
                            style=0.7 * rand((len(ids), 64), seed + 5), pn=rand((R, 128), seed + 6), sn=rand((R * 75,), seed + 7),
                            ph=rand((1,), seed + 8, normal=False), rows=R))
    my_rows = sum(b["rows"] for b in batches)
    audio = torch.empty(my_rows * 75, dtype=torch.float32, device=device)
    col = WaveformCollector([75 * f for f in frames], device, dst=0) if world > 1 else None

    def step():
        off = 0
        for b in batches:
            model.frame_path(b["seg"], b["asr"], b["pitch"], b["energy"], b["style"], b["pn"], b["sn"], b["ph"], batch_scope=False,
                             out=audio[75 * off : 75 * (off + b["rows"])])
            off += b["rows"]
        if col is not None:
            col.collect(audio)

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
    rows_all = [sum(frames[i] for i in p) for p in parts]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(audio).all())
    if rank == 0:
        audio_seconds = args.steps * sum(frames) * 75 / SR
        print(json.dumps({
            "metric": "utterances_per_sec", "value": round(args.steps * n_utt / elapsed, 3), "unit": "utt/s",
            "rtf": elapsed / audio_seconds, "realtime_x": audio_seconds / elapsed, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": precision,
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {wl['note']} ({sum(frames) * 75 / SR:.0f} s of audio per step), {precision} "
                            "Decoder + PriorEncoder/reverse flow + freegan iSTFT vocoder (stts_frame_path); one fixed global batch per step",
                "global_batch": n_utt, "utterances_per_rank": [len(p) for p in parts], "frames_per_rank": rows_all,
                "imbalance_max_over_mean": round(max(rows_all) / (sum(rows_all) / world), 4), "calls_per_rank_step": len(calls),
                "parallelism": (f"utterance-sharded x{world} (longest-first greedy on frame counts); RCCL broadcast(weights, once) + exact-size "
                                "point-to-point collection of the waveforms on rank 0 per step") if world > 1 else "single GPU",
            },
        }), flush=True)
    model.close()


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
    ap.add_argument("--cpu-utts", type=int, default=16, help="utterances the CPU baseline times (bounded sample: ~10-20 s of host work)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the counter passes run this script with the legs off
    ap.add_argument("--workload", choices=["cfg2", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2 (default, the bench line): 8 x 3 s per GPU, weak scaling; cfg4 / cfg5: BASELINE's sharded configs, one fixed global batch "
                         "partitioned over the ranks (strong scaling)")
    ap.add_argument("--utterances", type=int, default=0, help="cfg4 / cfg5: override the global batch size (rehearsals on one GPU)")
    args = ap.parse_args()
    global BATCH, T_MEL, T4
    BATCH, T_MEL = args.batch, args.mel_frames
    T4 = 4 * T_MEL

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    traffic, traffic_note = None, "not measured (--no-traffic / multi-GPU run)"
    if world == 1 and not args.pmc_child and not args.no_traffic and args.workload == "cfg2":
        # before anything here touches the GPU: the counter passes are child processes of a GPU-free parent
        traffic, traffic_note = measure_traffic(["--batch", str(args.batch), "--mel-frames", str(args.mel_frames), "--precision", args.precision])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    import torch.distributed as dist

    if os.environ.get("STTS_BENCH_ONE_GPU"):  # rehearsal of the N > 1 code path on a single-GPU box: every rank on cuda:0
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("STTS_BENCH_BACKEND", "nccl")  # "gloo": rehearsals without RCCL (one GPU shared by the ranks)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.workload != "cfg2":
        run_sharded(args, world, rank, local, device, dist)
        if world > 1:
            dist.destroy_process_group()
        return

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

    if rank == 0 and not args.pmc_child:
        # ---- roofline leg: HIP events around every launch of the step (same stream), 3 steps right after the timed region
        # (rank 0's own frame path, without the collection of the waveforms: the other ranks wait at the barrier below)
        lib = _lib.load()
        psteps = 3
        lib.stts_profile_begin()
        for _ in range(psteps):
            model.frame_path(seg, inp["asr"], inp["pitch"], inp["energy"], inp["style"], inp["prior_noise"], inp["src_noise"], inp["init_phase"],
                             batch_scope=True, out=audio)
        buf = C.create_string_buffer(1 << 16)
        _lib.check(lib.stts_profile_report(C.c_void_p(torch.cuda.current_stream().cuda_stream), buf, len(buf)))
        recs = json.loads(buf.value.decode())
        peak = MFMA_PEAK_TFLOPS[args.precision]
        con = [r for r in recs if r["kind"] == "contraction"]
        oth = [r for r in recs if r["kind"] == "other"]
        c_ms, c_fl, c_ex, c_n = (sum(r[k] for r in con) for k in ("ms", "gflop", "executed_gflop", "launches"))
        achieved = c_fl / c_ms if c_ms > 0 else 0.0        # GFLOP / ms = TFLOP/s
        executed = c_ex / c_ms if c_ms > 0 else 0.0
        all_n, all_ms = sum(r["launches"] for r in recs), sum(r["ms"] for r in recs)
        out["roofline"] = {
            "kernel": "conv_gemm_f32 + Winograd-form convs (timed with their transforms) + fused WaveNet-layer kernel = all Conv1d/Linear contractions of the step",
            "bound": "mfma",
            "achieved": round(achieved, 2),
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4),
            "achieved_note": "algorithmic (direct-conv) flops / kernel time: SURVEY.md 8d; a Winograd-form conv is credited with the flops of the direct convolution it replaces",
            "executed_tflops": round(executed, 2),
            "frac_executed": round(executed / peak, 4),
            "executed_note": "flops the matrix cores actually execute (F(6,7)/F(6,3)/F(2,5)/F(4,5) forms do n/(m r) of the direct multiplies) / the same kernel time",
            "traffic": traffic,
            "traffic_note": traffic_note,
            "launches_per_step": c_n // psteps,
            "avg_launch_ms": round(c_ms / max(c_n, 1), 5),
            "algorithmic_gflop_per_step": round(c_fl / psteps, 2),
            "executed_gflop_per_step": round(c_ex / psteps, 2),
            "gemm_ms_per_step": round(c_ms / psteps, 4),
            "gemm_share_of_step": round((c_ms / psteps) / (1e3 * elapsed / args.steps), 4),
            "contraction_kernels": [
                {"kernel": r["kernel"], "launches_per_step": r["launches"] // psteps, "ms_per_step": round(r["ms"] / psteps, 4),
                 "avg_us": round(1e3 * r["ms"] / r["launches"], 2), "tflops": round(r["gflop"] / r["ms"], 1), "executed_tflops": round(r["executed_gflop"] / r["ms"], 1)}
                for r in con],
            "hbm_kernels": [
                {"kernel": r["kernel"], "launches_per_step": r["launches"] // psteps, "avg_us": round(1e3 * r["ms"] / r["launches"], 2),
                 "algorithmic_mb_per_launch": round(r["mbytes"] / r["launches"], 3), "gb_per_s": round(r["mbytes"] / r["ms"], 1),
                 "frac_of_8tb_s": round(r["mbytes"] / r["ms"] / 8000.0, 4)}
                for r in sorted(oth, key=lambda r: -r["ms"])],
            "hbm_kernels_note": "bandwidth-/latency-bound kernels: algorithmic HBM bytes (SURVEY.md 8d: inputs read once + outputs written once) / event-timed duration, against 8 TB/s",
            "all_launches_per_step": all_n // psteps,
            "all_kernel_ms_per_step": round(all_ms / psteps, 4),
            "us_per_launch": round(1e3 * all_ms / max(all_n, 1), 2),
        }
        # ---- CPU baseline leg: the oracle (validated against reference goldens) on the host cores (N = 1 only)
        if world == 1 and not args.no_cpu_baseline:
            from oracle import stylish_oracle as O

            h = inp["host"]
            per_call = min(BATCH, max(1, args.cpu_utts))  # the workload's own batch per oracle call: BLAS sees the same matrix shapes the GPU step does
            calls = max(1, args.cpu_utts // per_call)
            n_utts = calls * per_call
            asr_b = np.ascontiguousarray(h["asr"][: per_call * T4].reshape(per_call, T4, -1).transpose(0, 2, 1))
            pitch_b, energy_b = h["pitch"][: per_call * T4].reshape(per_call, T4), h["energy"][: per_call * T4].reshape(per_call, T4)
            nzb = dict(prior_noise=h["nz"]["prior_noise"][:per_call], src_noise=h["nz"]["src_noise"][:per_call], init_phase=h["nz"]["init_phase"])
            t1 = time.perf_counter()
            for _ in range(calls):
                O.frame_path(asr_b, pitch_b, energy_b, h["style"][:per_call], nzb, sd)
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
                "sample": f"{n_utts} utterances of the same workload (3.0 s each, {per_call} per oracle call like the GPU step, {calls} calls), numpy oracle; BLAS pool = {blas_threads} threads "
                          f"of {os.cpu_count()} logical cores",
                "reference_torch_cpu_note": "survey container, 8 vCPU, reference torch-CPU code: 1.4 utt/s at B=1, 1.9-2.5 utt/s at B=8 (BASELINE.md §2)",
            }
    if rank == 0 and not args.pmc_child:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # rank 0's roofline leg is over
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
