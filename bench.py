#!/usr/bin/env python3
"""bench.py — utterances/s and RTF of the Stylish-TTS inference hot path on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

The bench line (`value`): BASELINE cfg2 — one step = one pass of `stts_frame_path` (Decoder → PriorEncoder + reverse flow +
post_flow → harmonic source → STFT → vocoder body → iSTFT → tanh) over 8 synthetic LJSpeech-shaped utterances of 3.0 s per GPU
(T = 240 mel frames, T4 = 960 vocoder frames, 72 000 samples) in fp32, inputs resident in HBM; weak scaling over the GPUs.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  Under `python -m torch.distributed.run` the ranks are given;
invoked plainly as `python bench.py --gpus N` this process is a GPU-free PARENT that compiles the library, starts the N ranks as
child processes (stylish_tts_amd/launcher.py), relays rank 0's JSON line and exits with the ranks' status.  Rank 0 broadcasts the
weights once; per step every rank runs its shard and the waveforms are collected on rank 0 (exact-size point-to-point transfers)
inside the timed region.  Rank 0 prints ONE JSON line.

Besides the bench line the same JSON carries (skipped with --no-legs):
  * `legs.cfg4_strong` / `legs.cfg5_strong`: BASELINE configs[3] / [4] as ONE fixed global batch (256 utterances of 0.25-10 s in fp32;
    512 segments of 10 s with fp16 operands) partitioned over the ranks by frame count — the strong-scaling numbers;
  * `legs.cfg3_full_chain`: BASELINE configs[2] — tokens → waveform through the whole chain (duration + pitch/energy predictors in
    fp32, frame path with bf16 operands) at 64 utterances of 50 tokens per GPU, with the phoneme-rate / frame-rate split;
  * `roofline`: HIP start/stop events on EVERY launch of the step in a profiling pass right after the timed region
    (`stts_profile_begin/report`): the dense contractions against the MFMA roof in algorithmic (direct-conv) flops AND in executed
    flops, calibrated against the un-instrumented step (event pairs on every launch stretch the step by some per cent); the
    bandwidth-bound kernels against 8 TB/s with their algorithmic bytes; `traffic`: HBM-side bytes per contraction
    launch from two `rocprofv3 --pmc` child passes that run BEFORE this process touches the GPU;
  * `cpu_baseline`: the numpy oracle on the host cores, a pool of worker processes over utterances (N = 1, before the GPU is touched).
`--workload cfg3 | cfg4 | cfg5` makes that configuration the (only) timed workload of the run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 24000
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_PEAK_TFLOPS = {"f32": FP32_MFMA_PEAK_TFLOPS, "f32_native": FP32_MFMA_PEAK_TFLOPS, "bf16": 2500.0, "f16": 2500.0}  # dense 16-bit MFMA (same guide)
# split-fp32 contractions (csrc/gemm.hip.h PREC_X3): an fp32 product = six bf16 x bf16 products on the bf16 matrix cores, so the
# speed of light of that form is the dense bf16 peak / 6, in fp32 multiply-add flops
X3_PRODUCTS = 6
X3_PEAK_TFLOPS = MFMA_PEAK_TFLOPS["bf16"] / X3_PRODUCTS
FRAME_MFLOP = 74.1           # frame-rate part, MFLOP per hop-75 frame
PHONEME_GFLOP_PER_TOKEN = 3.6 / 50.0  # ~3.6 GFLOP per 50-token utterance (text encoders, duration head, pitch/energy)
CONTRACTION_KERNELS = ("conv_gemm", "wn_fused", "wn_layer", "winograd_", "gemm16")
MAX_ROWS_PER_CALL = 220_000  # frames per stts_frame_path call (~12 GB of workspace; 64 x 10 s = 204 800 is the size the tests pin)

WORKLOADS = {
    # BASELINE.json configs[3]: 256 utterances of 0.25-10 s (seeded), fp32, one fixed global batch sharded over the ranks
    "cfg4": dict(n_utt=256, seconds=(0.25, 10.0), precision="f32", note="mixed-length batch: 256 utterances of 0.25-10 s"),
    # BASELINE.json configs[4]: 512 segments of 10 s, fp16 operands (the speaker-id input does not exist in the reference: SURVEY.md 8c)
    "cfg5": dict(n_utt=512, seconds=(10.0, 10.0), precision="f16", note="long-form batch: 512 segments of 10 s (single-speaker graph: the reference has no speaker-id embedding)"),
}


# ------------------------------------------------------------------------------------------------ CPU baseline (no GPU, worker pool)
def _cpu_worker_init(threads: int):
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = str(threads)
    global _W
    import numpy as np  # noqa: F401

    from stylish_tts_amd import params
    from stylish_tts_amd.config import load_model_config

    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(threads)
    except Exception:
        pass
    _W = params.synth_state_dict(params.module_spec("speech_predictor", load_model_config()), 0, prefix="speech_predictor.")


def _cpu_worker_run(job):
    """One utterance of the bench workload through the oracle's frame path (B = 1, the way the reference's callers run it)."""
    import numpy as np

    from oracle import stylish_oracle as O
    from stylish_tts_amd import synth

    b, t4 = job
    if b < 0:
        return 0.0  # start-up barrier: the worker has imported everything and built its weights
    tag = "bench.r0"  # utterance b of rank 0's cfg2 batch (cfg2_inputs)
    asr = synth.normal(f"{tag}.asr{b}", (t4, 128)).T[None].copy()
    pitch = synth.pitch_curve(f"{tag}.pitch{b}", 1, t4)
    energy = (synth.uniform(f"{tag}.energy{b}", (t4,)) * 2 + 2).astype(np.float32)[None]
    style = (synth.normal(f"{tag}.style", (8, 64)) * 0.7).astype(np.float32)[b : b + 1]
    nz8 = synth.path_noise(tag, 8, t4)
    nz = dict(prior_noise=nz8["prior_noise"][b : b + 1], src_noise=nz8["src_noise"][b : b + 1], init_phase=nz8["init_phase"])
    t0 = time.perf_counter()
    O.frame_path(asr, pitch, energy, style, nz, _W)
    return time.perf_counter() - t0


def cpu_baseline(args, t4: int):
    """The oracle (kind "port") on the host cores: `workers` processes x `threads` BLAS threads, utterances dealt over the workers.
    Runs before this process touches the GPU (the workers are child processes)."""
    import multiprocessing as mp

    cores = os.cpu_count() or 1
    share = min(cores, 16)  # the CPU share of a one-GPU box
    threads = max(1, min(args.cpu_threads, share))
    workers = max(1, min(args.cpu_workers or share // threads, args.cpu_utts))
    n_utts = max(workers, args.cpu_utts // workers * workers)
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers, initializer=_cpu_worker_init, initargs=(threads,)) as pool:
        pool.map(_cpu_worker_run, [(-1, t4)] * workers, chunksize=1)  # every worker is up
        t0 = time.perf_counter()
        per = pool.map(_cpu_worker_run, [(b % 8, t4) for b in range(n_utts)], chunksize=1)
        wall = time.perf_counter() - t0
    return {
        "value": round(n_utts / wall, 4), "unit": "utt/s", "cores": workers * threads, "kind": "port",
        "sample": f"{n_utts} utterances of the bench workload ({t4 * 75 / SR:.1f} s each) through the numpy oracle's frame path, one utterance per call like the "
                  f"reference's callers; {workers} worker processes x {threads} BLAS threads = {workers * threads} of the host's {cores} logical cores; "
                  f"{wall:.1f} s of wall time, {sum(per) / len(per):.2f} s per utterance per worker",
        "reference_torch_cpu_note": "survey container, 8 vCPU, the reference's own torch-CPU code: 1.4 utt/s at B=1, 1.9-2.5 utt/s at B=8 (BASELINE.md §2)",
    }


# ------------------------------------------------------------------------------------------------ HBM traffic (rocprofv3 --pmc children)
def measure_traffic(extra_args):
    """HBM-side bytes per contraction launch: two separate `rocprofv3 --pmc` passes over `bench.py --pmc-child` (child
    processes, started before this process initialises the GPU; the library is already compiled, the children never compile).
    Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KB, and
    FETCH_SIZE tallies 128-byte requests at 64 bytes (x2).  Returns (bytes_per_launch or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

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


# ------------------------------------------------------------------------------------------------ distributed context
class Ctx:
    def __init__(self, world, rank, local):
        self.world, self.rank, self.local = world, rank, local
        self.dist = None
        self.device = None
        self.backend = None

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, v: float) -> float:
        if self.world == 1:
            return v
        import torch

        t = torch.tensor([v], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_objects(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


def timed_steps(ctx: Ctx, step, steps: int, warmup: int, after_warmup=None) -> float:
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; max over ranks (seconds)."""
    import torch

    for _ in range(warmup):
        step()
    if after_warmup:
        after_warmup()
    ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ctx.local_elapsed = time.perf_counter() - t0  # this rank's own time (reported per rank next to the max)
    ctx.barrier()
    return ctx.max_over_ranks(time.perf_counter() - t0)


def device_identity(device) -> str:
    import torch

    p = torch.cuda.get_device_properties(device)
    ident = getattr(p, "uuid", None)
    return f"{p.name} #{device.index} {ident}" if ident is not None else f"{p.name} #{device.index}"


def broadcast_weights(ctx: Ctx, names, cfg):
    """{module: state dict} on every rank: rank 0 synthesizes, one flat fp32 broadcast per module (RCCL), once, outside the timed region."""
    from stylish_tts_amd import params
    from stylish_tts_amd.sharding import broadcast_state_dict

    out = {}
    for m in names:
        spec = params.module_spec(m, cfg)
        sd = params.synth_state_dict(spec, 0, prefix=m + ".") if ctx.rank == 0 else None
        if ctx.world > 1:
            sd = broadcast_state_dict(sd, spec, ctx.device if ctx.backend == "nccl" else "cpu", src=0)
        out[m] = sd
    return out


# ------------------------------------------------------------------------------------------------ cfg2: the bench line
def cfg2_inputs(rank: int, device, batch: int, t4: int):
    import numpy as np
    import torch

    from stylish_tts_amd import synth

    tag = f"bench.r{rank}"
    asr = np.concatenate([synth.normal(f"{tag}.asr{b}", (t4, 128)) for b in range(batch)])
    pitch = np.concatenate([synth.pitch_curve(f"{tag}.pitch{b}", 1, t4)[0] for b in range(batch)])
    energy = np.concatenate([(synth.uniform(f"{tag}.energy{b}", (t4,)) * 2 + 2).astype(np.float32) for b in range(batch)])
    style = (synth.normal(f"{tag}.style", (batch, 64)) * 0.7).astype(np.float32)
    nz = synth.path_noise(tag, batch, t4)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)  # noqa: E731
    return dict(
        asr=d(asr), pitch=d(pitch), energy=d(energy), style=d(style),
        prior_noise=d(nz["prior_noise"].transpose(0, 2, 1).reshape(batch * t4, 128)),
        src_noise=d(nz["src_noise"].reshape(-1)), init_phase=d(nz["init_phase"].reshape(-1)),
        host=dict(asr=asr, pitch=pitch, energy=energy, style=style, nz=nz),
    )


def run_cfg2(ctx: Ctx, args, traffic, traffic_note):
    import torch

    from stylish_tts_amd.config import load_model_config
    from stylish_tts_amd.runtime import HipModel, Segments
    from stylish_tts_amd.sharding import WaveformCollector

    B, t4 = args.batch, 4 * args.mel_frames
    cfg = load_model_config()
    w = broadcast_weights(ctx, ["speech_predictor"], cfg)
    model = HipModel(cfg, ctx.local, precision=args.precision)
    model.load_weights(w, which=7)
    seg = Segments([t4] * B, ctx.device)
    inp = cfg2_inputs(ctx.rank, ctx.device, B, t4)
    audio = torch.empty(B * t4 * 75, dtype=torch.float32, device=ctx.device)
    # global batch = world x B utterances; rank r owns utterances [r B, (r + 1) B): the collector is told that partition
    col = WaveformCollector([75 * t4] * (B * ctx.world), ctx.device, dst=0, parts=[list(range(r * B, (r + 1) * B)) for r in range(ctx.world)]) if ctx.world > 1 else None
    # N > 1: two audio buffers; step i's waveforms travel to rank 0 on the collector's own stream while step i + 1 computes (a buffer is
    # written again only after the transfer that read it has completed: its `done` event)
    bufs = [audio, torch.empty_like(audio)] if col is not None else [audio]
    done = [None, None]
    k = [0]

    def step():
        b = k[0] % len(bufs)
        k[0] += 1
        if done[b] is not None:
            torch.cuda.current_stream().wait_event(done[b])
        model.frame_path(seg, inp["asr"], inp["pitch"], inp["energy"], inp["style"], inp["prior_noise"], inp["src_noise"], inp["init_phase"],
                         batch_scope=True, out=bufs[b])
        if col is not None:
            done[b] = col.collect_async(bufs[b], timed=True)  # waveforms to rank 0 over xGMI, inside the timed region (the final synchronize waits for the last one)

    elapsed = timed_steps(ctx, step, args.steps, args.warmup, after_warmup=model.check_status)
    assert all(bool(torch.isfinite(a).all()) for a in bufs)
    step_ms_per_rank = ctx.gather_objects(round(1e3 * ctx.local_elapsed / args.steps, 4)) if hasattr(ctx, "local_elapsed") else None
    collect_ms = col.collect_ms() if col is not None else None
    utts = ctx.world * B * args.steps
    audio_seconds = utts * (t4 * 75 / SR)
    is_cfg2 = (B, args.mel_frames) == (8, 240) and args.precision in ("f32", "f32_native")
    devices = ctx.gather_objects(device_identity(ctx.device))
    out = {
        "metric": "utterances_per_sec", "value": round(utts / elapsed, 3), "unit": "utt/s",
        "rtf": elapsed / audio_seconds, "realtime_x": audio_seconds / elapsed,
        "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "f32_native" else args.precision, "data": "synthetic",
        "arithmetic": {"f32": "fp32: every activation and weight is the exact sum of three bf16 terms; the frame-rate contractions of the decoder and the vocoder form each fp32 product from the six "
                              "largest bf16 x bf16 cross terms on the bf16 matrix cores, fp32 accumulate (nothing is rounded to 16 bits; error against float64 at or below the f32 matrix cores', "
                              "tests/test_hip_split_fp32.py); the flow, norms, gates, FFTs: fp32 on v_mfma_f32 / VALU.  legs.cfg2_f32_matrix_cores = the same step with v_mfma_f32 only",
                       "f32_native": "fp32 on the f32 matrix cores (v_mfma_f32_32x32x2_f32) for every contraction"}.get(args.precision, f"{args.precision} matrix-core operands, fp32 accumulate"),
        "config": {
            "workload": ("cfg2: " if is_cfg2 else "side experiment: ") + f"LJSpeech-shaped batch={B} x {args.mel_frames / 80:.1f} s per GPU (T={args.mel_frames} mel frames, {t4 * 75} samples @24 kHz) "
                        + ("fp32" if args.precision in ("f32", "f32_native") else f"{args.precision} matrix-core operands, fp32 accumulate")
                        + ", Decoder + PriorEncoder/reverse flow + freegan iSTFT vocoder (stts_frame_path); no diffusion step exists in the reference",
            "batch_per_gpu": B, "global_batch": B * ctx.world, "frames_per_utt": t4,
            "parallelism": (f"utterance-sharded x{ctx.world}: {ctx.backend} broadcast(weights, once) + exact-size point-to-point collection of the waveforms on rank 0 per step"
                            if ctx.world > 1 else "single GPU"),
            "ranks": ctx.world if ctx.world == 1 else ctx.dist.get_world_size(), "backend": ctx.backend or "none", "devices": devices,
            "distinct_devices": len(set(devices)),
            "frames_per_rank": [B * t4] * ctx.world, "imbalance_max_over_mean": 1.0,
            "collect_bytes_per_step": 0 if ctx.world == 1 else 4 * 75 * t4 * B * (ctx.world - 1),
            "collect_ms_per_step": None if collect_ms is None else round(collect_ms, 4),  # on the collector's stream, overlapped with the next step's compute
            "ms_per_step_per_rank": step_ms_per_rank,  # every rank's own wall time over the timed steps (value uses the MAX)
        },
    }
    if not args.pmc_child:
        out["roofline"] = roofline_leg(ctx, args, model, lambda: model.frame_path(seg, inp["asr"], inp["pitch"], inp["energy"], inp["style"], inp["prior_noise"],
                                                                                 inp["src_noise"], inp["init_phase"], batch_scope=True, out=audio),
                                       1e3 * elapsed / args.steps, traffic, traffic_note)
    model.close()
    return out


def run_cfg2_native(ctx: Ctx, args):
    """The bench line's workload with every fp32 contraction on the f32 matrix cores (v_mfma_f32_32x32x2_f32: HipModel(precision="f32_native"), the
    path of rounds 1-3), same inputs, same step count: what the split-fp32 form of the contractions buys, measured in the same process."""
    import torch

    from stylish_tts_amd.config import load_model_config
    from stylish_tts_amd.runtime import HipModel, Segments

    B, t4 = args.batch, 4 * args.mel_frames
    cfg = load_model_config()
    w = broadcast_weights(ctx, ["speech_predictor"], cfg)
    model = HipModel(cfg, ctx.local, precision="f32_native")
    model.load_weights(w, which=7)
    seg = Segments([t4] * B, ctx.device)
    inp = cfg2_inputs(ctx.rank, ctx.device, B, t4)
    audio = torch.empty(B * t4 * 75, dtype=torch.float32, device=ctx.device)

    def step():
        model.frame_path(seg, inp["asr"], inp["pitch"], inp["energy"], inp["style"], inp["prior_noise"], inp["src_noise"], inp["init_phase"], batch_scope=True, out=audio)

    elapsed = timed_steps(ctx, step, args.steps, args.warmup, after_warmup=model.check_status)
    model.close()
    return {"metric": "utterances_per_sec", "value": round(ctx.world * B * args.steps / elapsed, 3), "unit": "utt/s", "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "dtype": "f32", "arithmetic": "v_mfma_f32_32x32x2_f32 / 16x16x4 for every contraction (no split form)", "steps": args.steps, "warmup": args.warmup}


def roofline_leg(ctx: Ctx, args, model, frame_step, ms_per_step, traffic, traffic_note, precision=None):
    """HIP events around every launch of `frame_step` (same stream), 3 steps right after the timed region; every rank measures its own
    GPU, rank 0's figures are reported with the spread over ranks.  Calibration: event pairs on every launch serialise the dispatches
    and stretch the sum of the kernel durations beyond the un-instrumented step, so every duration is scaled by
    (GPU time of the un-instrumented step) / (sum of the event-timed durations) when that ratio is below 1."""
    import ctypes as C

    import torch

    from stylish_tts_amd import _lib

    precision = precision or args.precision
    lib = _lib.load()
    psteps = 3
    # GPU time of the plain step (one event pair around three whole steps, same stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    frame_step()
    e0.record()
    for _ in range(psteps):
        frame_step()
    e1.record()
    torch.cuda.synchronize()
    plain_ms = e0.elapsed_time(e1) / psteps
    lib.stts_profile_begin()
    for _ in range(psteps):
        frame_step()
    buf = C.create_string_buffer(1 << 17)
    _lib.check(lib.stts_profile_report(C.c_void_p(torch.cuda.current_stream().cuda_stream), buf, len(buf)))
    recs = json.loads(buf.value.decode())
    peak = MFMA_PEAK_TFLOPS[precision]
    con = [r for r in recs if r["kind"] == "contraction"]
    oth = [r for r in recs if r["kind"] == "other"]
    c_ms, c_fl, c_ex, c_n = (sum(r[k] for r in con) for k in ("ms", "gflop", "executed_gflop", "launches"))
    all_n, all_ms = sum(r["launches"] for r in recs), sum(r["ms"] for r in recs)
    raw_all = all_ms / psteps
    # calibration against the TIMED step (the number `ms_per_step` reports): the instrumented pass serialises the dispatches (an event pair on every
    # launch, side stream off), so its kernel durations can sum to more than the step they describe; they are scaled down to it, never up
    scale = min(1.0, ms_per_step / raw_all) if raw_all > 0 else 1.0
    x3 = [r for r in con if r["kernel"].endswith("_x3")]   # split-fp32 form: bf16 matrix cores
    nat = [r for r in con if not r["kernel"].endswith("_x3")]

    def fam(rs, pk):
        ms, fl, ex, n = (sum(r[k] for r in rs) for k in ("ms", "gflop", "executed_gflop", "launches"))
        ms *= scale
        return dict(launches_per_step=n // psteps, ms_per_step=round(ms / psteps, 4), algorithmic_tflops=round(fl / ms, 2) if ms else 0.0,
                    executed_tflops=round(ex / ms, 2) if ms else 0.0, peak=round(pk, 1), frac=round(fl / ms / pk, 4) if ms else 0.0,
                    frac_executed=round(ex / ms / pk, 4) if ms else 0.0)

    x3_dominant = precision == "f32" and sum(r["ms"] for r in x3) > sum(r["ms"] for r in nat)
    dom = x3 if x3_dominant else con
    d_ms, d_fl, d_ex, d_n = (sum(r[k] for r in dom) for k in ("ms", "gflop", "executed_gflop", "launches"))
    if x3_dominant:
        peak = X3_PEAK_TFLOPS
    achieved = d_fl / (d_ms * scale) if d_ms > 0 else 0.0  # GFLOP / ms = TFLOP/s
    executed = d_ex / (d_ms * scale) if d_ms > 0 else 0.0
    per_rank = ctx.gather_objects(round(c_ms * scale / psteps, 4))
    head = {
        "kernel": ("the split-fp32 contractions: conv_gemm_f32<..., PREC_X3> (every Conv1d / Linear of the decoder and the vocoder; Winograd-form convs timed with their "
                   "transforms) and wn_fused_x3_kernel (the flow's fused WaveNet layers)" + ("; contractions still on the f32 matrix cores are listed under f32_mfma_contractions" if nat else "")
                   if x3_dominant else
                   "all Conv1d / Linear contractions of the step (conv_gemm kernels, Winograd-form convs timed with their transforms, fused WaveNet-layer kernels)"),
        "bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
        "achieved_note": "algorithmic (direct-conv) fp32 flops / calibrated kernel time: SURVEY.md 8d; a Winograd-form conv is credited with the flops of the direct convolution it replaces",
        "peak_note": (f"fp32 products formed as {X3_PRODUCTS} bf16 x bf16 products on the bf16 matrix cores (operands split exactly into three bf16 terms, fp32 accumulate): "
                      f"dense bf16 peak {MFMA_PEAK_TFLOPS['bf16']:.0f} TFLOP/s / {X3_PRODUCTS} = the fp32 multiply-add rate of this form; against the f32 matrix cores' "
                      f"{FP32_MFMA_PEAK_TFLOPS} TFLOP/s the same achieved figure reads {achieved / FP32_MFMA_PEAK_TFLOPS:.3f}") if x3_dominant else f"dense MFMA peak for {precision} operands",
        "executed_tflops": round(executed, 2), "frac_executed": round(executed / peak, 4),
        "executed_note": ("fp32 multiply-add flops whose products the matrix cores actually form (F(6,7)/F(6,3) Winograd forms do n/(m r) of the direct multiplies) / the same kernel time; "
                          + (f"x {X3_PRODUCTS} = {executed * X3_PRODUCTS:.0f} TFLOP/s of bf16 MFMA work against {MFMA_PEAK_TFLOPS['bf16']:.0f}: the same fraction" if x3_dominant else "F(2,5)/F(4,5) in the flow")),
    }
    if x3_dominant:
        if nat:
            head["f32_mfma_contractions"] = dict(fam(nat, FP32_MFMA_PEAK_TFLOPS), kernels=sorted({r["kernel"] for r in nat}),
                                                 note="contractions on v_mfma_f32_*: launches without a split instantiation (small batches: the flow's 16-row F(1,5) kernel)")
        head["all_contractions"] = dict(ms_per_step=round(c_ms * scale / psteps, 4), algorithmic_tflops=round(c_fl / (c_ms * scale), 2), executed_tflops=round(c_ex / (c_ms * scale), 2),
                                        vs_f32_mfma_peak=round(c_fl / (c_ms * scale) / FP32_MFMA_PEAK_TFLOPS, 4))
    return {
        **head,
        "event_calibration": {"timed_step_ms": round(ms_per_step, 4), "plain_step_gpu_ms": round(plain_ms, 4), "sum_of_event_timed_kernels_ms": round(raw_all, 4), "scale": round(scale, 4),
                              "frac_uncalibrated": round(d_fl / d_ms / peak, 4) if d_ms > 0 else 0.0,
                              "note": "durations x scale, scale = min(1, timed step / sum of the event-timed kernel durations): the instrumented pass (an event pair on every launch, "
                                      "no side stream) cannot describe a step longer than the one that was timed; rocprofv3 traces the step WITH its side stream, where the kernels that "
                                      "overlap each other run 5-10 % longer each: its per-kernel durations (profiles/, tools/summarize_r04.py) sum to more than the step"},
        "traffic": traffic, "traffic_note": traffic_note,
        "launches_per_step": d_n // psteps, "avg_launch_ms": round(d_ms * scale / max(d_n, 1), 5),
        "algorithmic_gflop_per_step": round(c_fl / psteps, 2), "executed_gflop_per_step": round(c_ex / psteps, 2),
        "gemm_ms_per_step": round(c_ms * scale / psteps, 4), "gemm_ms_per_step_per_rank": per_rank,
        "gemm_share_of_step": round((c_ms * scale / psteps) / ms_per_step, 4),
        "contraction_kernels": [
            {"kernel": r["kernel"], "launches_per_step": r["launches"] // psteps, "ms_per_step": round(r["ms"] * scale / psteps, 4),
             "avg_us": round(1e3 * r["ms"] * scale / r["launches"], 2), "tflops": round(r["gflop"] / (r["ms"] * scale), 1),
             "executed_tflops": round(r["executed_gflop"] / (r["ms"] * scale), 1)}
            for r in con],
        "hbm_kernels": [
            {"kernel": r["kernel"], "launches_per_step": r["launches"] // psteps, "avg_us": round(1e3 * r["ms"] * scale / r["launches"], 2),
             "algorithmic_mb_per_launch": round(r["mbytes"] / r["launches"], 3), "gb_per_s": round(r["mbytes"] / (r["ms"] * scale), 1),
             "frac_of_8tb_s": round(r["mbytes"] / (r["ms"] * scale) / 8000.0, 4)}
            for r in sorted(oth, key=lambda r: -r["ms"])],
        "hbm_kernels_note": "bandwidth-/latency-bound kernels: algorithmic HBM bytes (SURVEY.md 8d: inputs read once + outputs written once) / calibrated duration, against 8 TB/s",
        "all_launches_per_step": all_n // psteps, "all_kernel_ms_per_step": round(raw_all * scale, 4),
        "us_per_launch": round(1e3 * raw_all * scale / max(all_n // psteps, 1), 2),
    }


# ------------------------------------------------------------------------------------------------ cfg4 / cfg5: one fixed global batch, sharded
def run_sharded(ctx: Ctx, args, name: str, steps: int, warmup: int, n_utt: int = 0):
    """ONE fixed global batch, utterances partitioned over the ranks by frame count (longest first), every rank runs its shard through
    stts_frame_path (in calls of at most MAX_ROWS_PER_CALL frames) and the waveforms are collected on rank 0 with exact sizes
    (sharding.WaveformCollector) inside the timed step.  Strong scaling: `value` = utterances of the global batch / step time."""
    import numpy as np
    import torch

    from stylish_tts_amd.config import load_model_config
    from stylish_tts_amd.runtime import HipModel, Segments
    from stylish_tts_amd.sharding import WaveformCollector, partition_utterances

    wl = WORKLOADS[name]
    precision = wl["precision"] if args.precision == "f32" else args.precision
    cfg = load_model_config()
    w = broadcast_weights(ctx, ["speech_predictor"], cfg)
    model = HipModel(cfg, ctx.local, precision=precision)
    model.load_weights(w, which=7)
    device = ctx.device
    n_utt = n_utt or args.utterances or wl["n_utt"]
    rng = np.random.default_rng(4)
    secs = rng.uniform(wl["seconds"][0], wl["seconds"][1], n_utt)
    frames = [int(4 * max(1, round(80 * v))) for v in secs]  # vocoder frames per utterance (T4 = 4 T)
    parts = partition_utterances(frames, ctx.world)
    mine = parts[ctx.rank]
    calls, cur, rows = [], [], 0  # this rank's calls: consecutive utterances of its shard, at most MAX_ROWS_PER_CALL frames each
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
        seed = 1000 * ctx.rank + ci
        pitch = 80.0 + 220.0 * rand((R,), seed + 1, normal=False)
        pitch = torch.where(rand((R,), seed + 2, normal=False) < 0.3, torch.zeros_like(pitch), pitch)  # ~30 % unvoiced frames
        batches.append(dict(seg=Segments(L, device), asr=rand((R, 128), seed + 3), pitch=pitch, energy=2 + 2 * rand((R,), seed + 4, normal=False),
                            style=0.7 * rand((len(ids), 64), seed + 5), pn=rand((R, 128), seed + 6), sn=rand((R * 75,), seed + 7),
                            ph=rand((1,), seed + 8, normal=False), rows=R))
    my_rows = sum(b["rows"] for b in batches)
    audio = torch.empty(max(my_rows, 1) * 75, dtype=torch.float32, device=device)[: my_rows * 75]
    col = WaveformCollector([75 * f for f in frames], device, dst=0) if ctx.world > 1 else None

    def step():
        off = 0
        for b in batches:
            model.frame_path(b["seg"], b["asr"], b["pitch"], b["energy"], b["style"], b["pn"], b["sn"], b["ph"], batch_scope=False,
                             out=audio[75 * off: 75 * (off + b["rows"])])
            off += b["rows"]
        if col is not None:
            col.collect(audio)

    elapsed = timed_steps(ctx, step, steps, warmup, after_warmup=model.check_status)
    assert bool(torch.isfinite(audio).all())
    rows_all = [sum(frames[i] for i in p) for p in parts]
    audio_seconds = steps * sum(frames) * 75 / SR
    out = {
        "metric": "utterances_per_sec", "value": round(steps * n_utt / elapsed, 3), "unit": "utt/s",
        "rtf": elapsed / audio_seconds, "realtime_x": audio_seconds / elapsed, "n_gpus": ctx.world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": precision,
        "data": "synthetic",
        "frame_tflops_algorithmic": round(sum(frames) * steps * FRAME_MFLOP * 1e-6 / elapsed, 1),
        "frame_tflops_note": f"SURVEY.md 8d: 74.1 MFLOP of direct-conv work per hop-75 frame / wall time of the whole step (every kernel, collection included), all {ctx.world} GPU(s); "
                             f"dense peak {MFMA_PEAK_TFLOPS[precision]} TFLOP/s per GPU for {precision} operands" + (" - the fp32 Winograd forms execute ~0.6 of these flops, so this is not a pipe utilisation" if precision == "f32" else ""),
        "config": {
            "workload": f"{name}: {wl['note']} ({sum(frames) * 75 / SR:.0f} s of audio per step), {precision} "
                        "Decoder + PriorEncoder/reverse flow + freegan iSTFT vocoder (stts_frame_path); one fixed global batch per step",
            "global_batch": n_utt, "utterances_per_rank": [len(p) for p in parts], "frames_per_rank": rows_all,
            "imbalance_max_over_mean": round(max(rows_all) / (sum(rows_all) / ctx.world), 4), "calls_per_rank_step": len(calls),
            "ranks": ctx.world if ctx.world == 1 else ctx.dist.get_world_size(), "backend": ctx.backend or "none",
            "collect_bytes_per_step": 0 if ctx.world == 1 else 4 * 75 * (sum(frames) - rows_all[0]),
            "parallelism": (f"utterance-sharded x{ctx.world} (longest-first greedy on frame counts); {ctx.backend} broadcast(weights, once) + exact-size "
                            "point-to-point collection of the waveforms on rank 0 per step") if ctx.world > 1 else "single GPU",
        },
    }
    model.close()
    return out


# ------------------------------------------------------------------------------------------------ cfg3: the full chain, tokens -> waveform
def cfg3_weights(ctx: Ctx, cfg):
    """Synthetic weights of the five inference modules; the duration head's bias is shaped so that the synthetic model predicts
    LJSpeech-like durations (~5 mel frames per token: 50 tokens -> T ~ 240 = 3 s) instead of the ~15 frames per
    token a near-uniform softmax over the 16-class table gives."""
    import numpy as np

    from stylish_tts_amd import params

    w = broadcast_weights(ctx, list(params.MODULE_SPECS), cfg)
    b = np.asarray(w["duration_predictor"]["duration_proj.linear_layer.bias"], np.float32).copy()
    table = np.array([1, 2, 3, 4, 5, 6, 7, 9, 12, 15, 18, 22, 27, 32, 38, 46], np.float32)
    b += (-1.5 * np.abs(table - 4.8)).astype(np.float32)  # classes 4 / 5 / 6 frames win the argmax, token-dependent logits pick among them
    w["duration_predictor"]["duration_proj.linear_layer.bias"] = b
    return w


def run_cfg3(ctx: Ctx, args, steps: int, warmup: int):
    """BASELINE configs[2]: 64 utterances x 50 tokens per GPU through the whole chain (Synthesizer: duration predictor ->
    DurationProcessor -> pitch/energy -> speech predictor), frame path with bf16 operands, phoneme-rate predictors in fp32.
    "Style diffusion" does not exist in the reference: the stochastic prior + reverse flow is what runs.  Weak scaling."""
    import torch

    from stylish_tts_amd import synth
    from stylish_tts_amd.config import load_model_config
    from stylish_tts_amd.pipeline import Synthesizer
    from stylish_tts_amd.runtime import HipModel

    precision = "bf16" if args.precision == "f32" else args.precision
    B, P = args.cfg3_batch, args.cfg3_tokens
    cfg = load_model_config()
    w = cfg3_weights(ctx, cfg)
    eng = HipModel(cfg, ctx.local, precision=precision)
    eng.load_weights(w, which=255)
    syn = Synthesizer(eng, adapt=True)  # (the synthetic weights predict ~22 frames per token, four times real speech: the capacity ratio settles during the warm-up calls)
    toks = [synth.tokens(f"bench.cfg3.r{ctx.rank}.{i}", 1, P, cfg.text_encoder.tokens)[0].tolist() for i in range(B)]
    waves, det = syn(toks, return_details=True)
    frames = det["frames"]  # mel frames per utterance
    R4 = 4 * sum(frames)
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(77 + ctx.rank)
    noise = dict(prior_noise=torch.randn(R4, 128, generator=gen, device=ctx.device), src_noise=torch.randn(R4 * 75, generator=gen, device=ctx.device),
                 init_phase=torch.rand(1, generator=gen, device=ctx.device))
    state = {}
    W = max(1, args.cfg3_workers)

    def step():
        state["waves"] = syn(toks, noise=noise)

    if W == 1:
        elapsed = timed_steps(ctx, step, steps, warmup)
    else:
        # W batches in flight (Synthesizer.map: each on its own stream from its own host thread): one batch's phoneme-rate stages - hundreds of small,
        # latency-bound launches - run beside another batch's frame path.  Still EXACTLY `steps` batches inside the timed region, every one of them whole.
        for _ in range(max(1, warmup // W)):
            syn.map([toks] * W, workers=W, noise=[noise] * W)
        ctx.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = syn.map([toks] * steps, workers=W, noise=[noise] * steps)
        torch.cuda.synchronize()
        ctx.barrier()
        elapsed = ctx.max_over_ranks(time.perf_counter() - t0)
        state["waves"] = outs[-1]
    assert all(bool(torch.isfinite(x).all()) for x in state["waves"])
    # stage split (one extra call, events on the caller's stream): phoneme-rate part = everything before the frame path
    syn.stage_times(toks, noise)  # (the caller's own lane may not have run yet when the batches went through map()'s lanes: its workspaces are sized by this call)
    t = syn.stage_times(toks, noise)
    utts = ctx.world * B * steps
    audio_seconds = ctx.world * steps * sum(frames) * 300 / SR
    ph_gflop = PHONEME_GFLOP_PER_TOKEN * P * B
    fr_gflop = FRAME_MFLOP * 1e-3 * R4
    out = {
        "metric": "utterances_per_sec", "value": round(utts / elapsed, 3), "unit": "utt/s",
        "rtf": elapsed / audio_seconds, "realtime_x": audio_seconds / elapsed, "n_gpus": ctx.world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": precision, "data": "synthetic",
        "config": {
            "workload": f"cfg3: LJSpeech-shaped batch={B} x {P} tokens per GPU, tokens -> waveform through the whole chain (duration predictor, DurationProcessor, pitch/energy "
                        f"predictor in fp32; Decoder + prior/reverse flow + vocoder with {precision} matrix-core operands, fp32 accumulate); the reference has no style-diffusion "
                        "step: the stochastic prior + reverse flow is what runs" + (f"; {W} batches in flight (Synthesizer.map: every batch whole, on its own stream / host thread; "
                        "per-batch latency is the one-batch figure, ~15.7 ms)" if W > 1 else ""),
            "batch_per_gpu": B, "tokens_per_utt": P, "mel_frames_per_utt_mean": round(sum(frames) / B, 1), "audio_seconds_per_utt_mean": round(sum(frames) * 300 / SR / B, 2),
            "duration_bias_note": "duration_proj bias shaped so the synthetic model predicts ~5 frames per token (SURVEY.md 8d: P = 50 -> T = 240)",
            "host_syncs_between_duration_and_frame_path": syn.host_syncs_per_call,
            "batches_in_flight": W, "capacity_retries": syn.capacity_retries,
        },
        "roofline": {
            "phoneme_rate": {"ms": round(t["phoneme_ms"], 3), "gflop": round(ph_gflop, 1), "tflops": round(ph_gflop / t["phoneme_ms"], 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                             "frac": round(ph_gflop / t["phoneme_ms"] / FP32_MFMA_PEAK_TFLOPS, 4), "dtype": "f32",
                             "note": "SURVEY.md 8d: ~3.6 GFLOP per 50-token utterance; latency-bound (hundreds of small launches)"},
            "frame_rate": {"ms": round(t["frame_ms"], 3), "gflop": round(fr_gflop, 1), "tflops": round(fr_gflop / t["frame_ms"], 1), "peak": MFMA_PEAK_TFLOPS[precision],
                           "frac": round(fr_gflop / t["frame_ms"] / MFMA_PEAK_TFLOPS[precision], 4), "dtype": precision,
                           "note": "SURVEY.md 8d: 74.1 MFLOP per hop-75 frame, whole stts_frame_path (contractions + bandwidth-bound kernels) against the dense 16-bit MFMA peak"},
            "stage_note": "one serialised call with events between the stages; in the timed steps the phoneme-rate side streams overlap",
        },
    }
    eng.close()
    return out


# ------------------------------------------------------------------------------------------------ N > 1 plumbing without a GPU (CPU test of the launcher)
def plumbing_check(args, world, rank):
    """tests/test_bench_launcher.py: the ranks started by the parent rendezvous over gloo on the CPU and run the sharded workload's
    partition + exact-size collection on fabricated waveforms (utterance i filled with i).  No GPU, no compute, no metric."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from stylish_tts_amd.sharding import WaveformCollector, partition_utterances

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if args.fail_rank == rank:
            raise SystemExit(3)
        rng = np.random.default_rng(4)
        frames = [int(4 * max(1, round(80 * v))) for v in rng.uniform(0.25, 2.0, 24)]
        parts = partition_utterances(frames, world)
        col = WaveformCollector([3 * f for f in frames], "cpu", dst=0)
        local = torch.cat([torch.full((3 * frames[i],), float(i)) for i in parts[rank]]) if parts[rank] else torch.zeros(0)
        col.collect(local)
        ok = True
        if rank == 0:
            ok = all(bool((col.utterance(i) == float(i)).all()) and col.utterance(i).numel() == 3 * frames[i] for i in range(len(frames)))
        t = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        cpus = [None] * world
        dist.all_gather_object(cpus, len(os.sched_getaffinity(0)))  # (pin_rank_cpus ran at rank start)
        if rank == 0:
            rows = [sum(frames[i] for i in p) for p in parts]
            print(json.dumps({"plumbing_check": True, "ok": bool(t.item() == 1.0), "metric": "utterances_per_sec", "value": None, "n_gpus": world,
                              "ranks": dist.get_world_size(), "backend": "gloo", "frames_per_rank": rows, "cpus_per_rank": cpus,
                              "imbalance_max_over_mean": round(max(rows) / (sum(rows) / world), 4)}), flush=True)
        if t.item() != 1.0:
            raise SystemExit(4)
    finally:
        dist.destroy_process_group()


def pin_rank_cpus(local_rank: int, local_world: int):
    """A rank's host threads (the launch thread, the two side-stream workers of the full chain, RCCL's proxy thread) stay on that rank's own
    contiguous slice of the cores this process may use - slice r of `local_world` equal slices, which on a two-socket host also keeps ranks
    0..N/2-1 on the first socket's cores (the GPUs are enumerated socket by socket).  STTS_NO_AFFINITY=1 leaves the scheduler alone."""
    if os.environ.get("STTS_NO_AFFINITY") or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    per = len(cpus) // max(local_world, 1)
    if per < 2:  # fewer than two cores per rank: pinning would only hurt
        return None
    mine = cpus[local_rank * per : (local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine


# ------------------------------------------------------------------------------------------------ main
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU per step (BASELINE cfg2 = 8; other values are side experiments)")
    ap.add_argument("--mel-frames", type=int, default=240, help="mel frames per utterance (cfg2 = 240 = 3.0 s)")
    ap.add_argument("--precision", choices=["f32", "f32_native", "bf16", "f16"], default="f32",
                    help="operand precision of the frame-rate contractions; f32 = BASELINE cfg2 (the bench line); cfg3 / cfg5 default to bf16 / f16")
    ap.add_argument("--cpu-utts", type=int, default=256, help="utterances the CPU baseline times (bounded sample: ~10-30 s of host work)")
    ap.add_argument("--cpu-threads", type=int, default=2, help="BLAS threads per CPU-baseline worker process")
    ap.add_argument("--cpu-workers", type=int, default=0, help="CPU-baseline worker processes (0: fill the one-GPU box's share of 16 cores)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--no-legs", action="store_true", help="only the bench line (no cfg3 / cfg4 / cfg5 legs)")
    ap.add_argument("--legs", default="f32mfma,cfg4,cfg5,cfg3", help="comma-separated legs run after the bench line")
    ap.add_argument("--leg-steps", type=int, default=5)
    ap.add_argument("--leg-warmup", type=int, default=2)
    ap.add_argument("--cfg3-batch", type=int, default=64)
    ap.add_argument("--cfg3-tokens", type=int, default=50)
    ap.add_argument("--cfg3-workers", type=int, default=4, help="cfg3: batches in flight (Synthesizer.map); 1 = one call after the other (measured: 1: 15.7 ms per batch, 2: 16.2, 3: 14.1, 4: 13.6)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the counter passes run this script with the legs off
    ap.add_argument("--plumbing-check", action="store_true", help=argparse.SUPPRESS)  # CPU test of the N > 1 launcher + collection (no GPU, no metric)
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2 (default, the bench line + legs): 8 x 3 s per GPU, weak scaling; cfg3: the full chain at 64 x 50 tokens per GPU; cfg4 / cfg5: BASELINE's "
                         "sharded configs, one fixed global batch partitioned over the ranks (strong scaling)")
    ap.add_argument("--utterances", type=int, default=0, help="cfg4 / cfg5: override the global batch size (rehearsals on one GPU)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    import __graft_entry__ as entry

    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.pmc_child:
        # ---- the driver's plain `python bench.py --gpus N`: this process is the GPU-free parent of N ranks
        from stylish_tts_amd.launcher import launch_ranks

        if not args.plumbing_check:
            entry.compile()  # hipcc here, once, before any rank exists (the ranks find a fresh library)
        rc, _ = launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, timeout=float(os.environ.get("STTS_BENCH_TIMEOUT", "1500")))
        sys.exit(rc)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        pin_rank_cpus(int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}; run `python bench.py --gpus N` (it starts its own ranks) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if args.plumbing_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        return plumbing_check(args, world, rank)

    # ---- nothing below this line may compile: build first, while this process has not touched the GPU
    if args.pmc_child:
        if entry.is_stale():  # under rocprofv3 the profiler's library has already initialised the GPU: never start hipcc from here
            raise SystemExit("bench.py --pmc-child: libstylish_hip.so is missing or older than its sources; the parent compiles, the counter passes never do")
    else:
        entry.compile()  # file-locked: with several ranks one compiles, the others wait
    cpu = None
    traffic, traffic_note = None, "not measured (--no-traffic / multi-GPU run / another workload)"
    if world == 1 and not args.pmc_child and args.workload == "cfg2":
        if not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(args, 4 * args.mel_frames)  # worker processes, before the GPU is touched
            except Exception as e:  # noqa: BLE001
                cpu = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if not args.no_traffic:
            traffic, traffic_note = measure_traffic(["--batch", str(args.batch), "--mel-frames", str(args.mel_frames), "--precision", args.precision])

    import torch
    import torch.distributed as dist

    entry.load()
    ctx = Ctx(world, rank, local)
    if os.environ.get("STTS_BENCH_ONE_GPU"):  # rehearsal of the N > 1 code path on a single-GPU box: every rank on cuda:0
        ctx.local = 0
    torch.cuda.set_device(ctx.local)
    ctx.device = torch.device("cuda", ctx.local)
    ctx.dist = dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ctx.backend = os.environ.get("STTS_BENCH_BACKEND", "nccl")  # "gloo": rehearsals without RCCL (one GPU shared by the ranks)
        if ctx.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=ctx.device)
        else:
            dist.init_process_group(ctx.backend, rank=rank, world_size=world)

    if args.workload == "cfg2":
        out = run_cfg2(ctx, args, traffic, traffic_note)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if not args.no_legs and not args.pmc_child and (args.batch, args.mel_frames, args.precision) == (8, 240, "f32"):
            legs = {}
            for leg in [x for x in args.legs.split(",") if x]:
                t0 = time.perf_counter()
                try:
                    if leg == "f32mfma":
                        legs["cfg2_f32_matrix_cores"] = run_cfg2_native(ctx, args)
                    elif leg in ("cfg4", "cfg5"):
                        legs[leg + "_strong"] = run_sharded(ctx, args, leg, args.leg_steps, args.leg_warmup)
                    elif leg == "cfg3":
                        legs["cfg3_full_chain"] = run_cfg3(ctx, args, args.leg_steps, args.leg_warmup)
                except Exception as e:  # noqa: BLE001 - a leg never takes the bench line down; all ranks fail alike (same inputs, same code)
                    legs[leg] = {"error": f"{type(e).__name__}: {e}"}
                if rank == 0:
                    print(f"[bench] leg {leg}: {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
            out["legs"] = legs
    elif args.workload == "cfg3":
        out = run_cfg3(ctx, args, args.steps, args.warmup)
    else:
        out = run_sharded(ctx, args, args.workload, args.steps, args.warmup)
    if rank == 0 and not args.pmc_child:
        print(json.dumps(out), flush=True)
    ctx.barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
