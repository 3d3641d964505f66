"""tokens → waveforms: the whole inference chain on packed, ragged batches.

The reference has no packaged inferencer (README roadmap): its callers loop per utterance over
DurationPredictor → DurationProcessor → ExportModel (``train/test_onnx.py:48-79``, ``train/stage_type.py:483-523``).
This is that composition for a list of utterances of any lengths in ONE pass: every stage runs on the packed
sequences (per-utterance semantics, no padding).

The reference reads the predicted durations on the host between its two models (``test_onnx.py:65-66``) because the
alignment matrix must be sized.  Here nothing is read back in the middle: the frame-rate stages run on CAPACITY SEGMENTS
(include/stylish_hip.h, STTS_SEG_CAPACITY) - every buffer and grid is sized by an upper bound of each utterance's frame
count, the real offsets are computed on the device from the durations (``stts_frame_offsets``) and the waveforms come out
packed by the real lengths, which the host reads ONCE, with the audio.  The upper bound is frames-per-token x tokens: it
is ``frames_per_token`` (default 8; the reference's table allows 46 per token, real speech averages 5-6) - a FIXED function of
the token counts by default, so the same text gets the same buffers, launch plans and bits in every call - and a call whose
prediction does not fit (the frame counts read with the audio exceed a capacity; the truncated segments kept every kernel in
bounds) is repeated with exactly the capacities it asked for.  ``adapt=True`` lets the ratio follow the model's predictions
in steps of 2 frames per token (fewer repeated calls for slow voices; capacities - and through the launch plans the last bits
of the waveform - can then change between two calls of the same text while the ratio is still settling).
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .runtime import CapacityOverflow, HipModel, Segments
from .text import TextCleaner, frame_tokens, to_int16, write_wav


class Synthesizer:
    """engine: a HipModel with all components finalized (load_weights(..., which=255))."""

    MAX_FRAMES_PER_TOKEN = 46  # the largest entry of the reference's duration table (train/utils.py:391-393)

    def __init__(self, engine: HipModel, frames_per_token: float = 8.0, adapt: bool = False):
        """frames_per_token: capacity (mel frames per token) the frame-rate buffers are sized by.  adapt=False (default): the same
        capacities for the same token counts in every call, hence the same launch plans and bit-identical results from call to call;
        adapt=True: the ratio follows what the model predicts, quantised to steps of 2 frames per token."""
        self.eng = engine
        self.cfg = engine.cfg
        self._ratio = float(frames_per_token)
        self._adapt = adapt
        self.capacity_retries = 0  # calls repeated because a prediction did not fit its capacity
        self.text_cleaner = TextCleaner(self.cfg.symbol) if "symbol" in self.cfg else None
        self._lane = self._new_lane()
        self._lanes: List[dict] = []  # extra lanes of map(): one per concurrent call

    def _new_lane(self, own_stream: bool = False) -> dict:
        # the three text encoders only share the tokens: two of them (with their style encoders) run on side streams
        # next to the duration predictor and the host read of the frame counts, each issued from its own host thread: a
        # text encoder is ~120 launches (~0.5 ms of host time), so one thread cannot feed three streams (the C calls
        # release the GIL)
        dev = self.eng.device
        return dict(side=[torch.cuda.Stream(device=dev) for _ in range(2)], pool=ThreadPoolExecutor(max_workers=2, thread_name_prefix="stts-side"),
                    main=torch.cuda.Stream(device=dev) if own_stream else None)

    def map(self, batches: Sequence[Sequence[Sequence[int]]], workers: int = 2, noise: Optional[Sequence[Optional[Dict[str, torch.Tensor]]]] = None):
        """Several batches, `workers` of them in flight: each call runs on its own stream from its own host thread, so one
        batch's phoneme-rate stages (hundreds of small launches, the host read of its frame counts) overlap another's frame
        path.  Results in input order; per batch the arithmetic is that of __call__."""
        if workers <= 1 or len(batches) <= 1:
            return [self(b, noise=None if noise is None else noise[i]) for i, b in enumerate(batches)]
        while len(self._lanes) < workers:
            self._lanes.append(self._new_lane(own_stream=True))
        caller = torch.cuda.current_stream(self.eng.device)
        fork = torch.cuda.Event()
        fork.record(caller)

        def run(i):
            lane = self._lanes[i % workers]
            torch.cuda.set_device(self.eng.device)
            with torch.cuda.stream(lane["main"]):
                lane["main"].wait_event(fork)
                out = self._run(batches[i], None if noise is None else noise[i], False, lane)
                for w in out:
                    w.record_stream(caller)
            return out

        if not hasattr(self, "_map_pool") or self._map_pool._max_workers < workers:
            self._map_pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="stts-lane")
        # a lane serves its batches in order (its stream and workspace are its own), different lanes run concurrently
        futs = [self._map_pool.submit(lambda k=k: [run(i) for i in range(k, len(batches), workers)]) for k in range(workers)]
        per_lane = [f.result() for f in futs]
        for lane in self._lanes[:workers]:
            caller.wait_stream(lane["main"])
        return [per_lane[i % workers][i // workers] for i in range(len(batches))]

    def infer(self, texts: Sequence[str], noise: Optional[Dict[str, torch.Tensor]] = None, out_prefix: Optional[str] = None, combine: bool = False):
        """Phoneme strings → int16 waveforms, all utterances in one pass (the loop body of ``train/test_onnx.py:48-90``).

        With ``out_prefix`` the samples are also written as ``<prefix>_<i>.wav`` (or ``<prefix>_combined.wav``).
        """
        if self.text_cleaner is None:
            raise ValueError("model config has no 'symbol' section")
        waves = self([frame_tokens(self.text_cleaner(t)) for t in texts], noise=noise)
        samples = [to_int16(w) for w in waves]
        if out_prefix is not None:
            if combine:
                write_wav(f"{out_prefix}_combined.wav", np.concatenate(samples, axis=-1), self.cfg.sample_rate)
            else:
                for i, smp in enumerate(samples):
                    write_wav(f"{out_prefix}_{i}.wav", smp, self.cfg.sample_rate)
        return samples

    def __call__(self, token_lists: Sequence[Sequence[int]], noise: Optional[Dict[str, torch.Tensor]] = None, return_details: bool = False):
        return self._run(token_lists, noise, return_details, self._lane)

    host_syncs_per_call = 0  # reads of device data by the host between the duration predictor and the frame path

    def stage_times(self, token_lists, noise=None) -> Dict[str, float]:
        """One call with events between the stages on the caller's stream: milliseconds of the phoneme-rate part (everything up to
        the frame path's inputs: three text encoders, styles, durations, pitch / energy, length regulator) and of the frame path."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        lane = dict(self._lane, events=ev)  # the events travel with this call only (concurrent map() lanes record nothing into them)
        self._run(token_lists, noise, False, lane)
        torch.cuda.synchronize(self.eng.device)
        return dict(phoneme_ms=ev[0].elapsed_time(ev[1]), frame_ms=ev[1].elapsed_time(ev[2]))

    def _capacities(self, L: Sequence[int]) -> List[int]:
        r = min(self._ratio, float(self.MAX_FRAMES_PER_TOKEN))
        return [min(self.MAX_FRAMES_PER_TOKEN * n, int(np.ceil(r * n)) + 8) for n in L]

    @torch.no_grad()
    def _run(self, token_lists, noise, return_details, lane):
        L = [len(t) for t in token_lists]
        caps = self._capacities(L)
        while True:
            try:
                return self._run_once(token_lists, L, caps, noise, return_details, lane)
            except CapacityOverflow as e:  # rare: the model predicted more frames than this call's buffers hold - repeat it with what it asked for
                self.capacity_retries += 1
                caps = [max(int(n), c) for n, c in zip(e.need, caps)]

    def _run_once(self, token_lists, L, caps, noise, return_details, lane):
        eng, dev = self.eng, self.eng.device
        toks = torch.tensor([int(v) for t in token_lists for v in t], dtype=torch.int64, device=dev)
        sp = Segments(L, dev)
        main = torch.cuda.current_stream(dev)
        ev = lane.get("events")
        if ev:
            ev[0].record(main)
        ready = torch.cuda.Event()
        ready.record(main)
        # 2a/3a. the pitch/energy and speech text + style encoders: phoneme-rate, independent of the durations
        def encode(which, stream):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(stream):
                stream.wait_event(ready)
                e = eng.text_encoder(which, sp, toks)
                y = eng.text_style(which, sp, e)
                for t in (e, y):
                    t.record_stream(main)  # consumed on the caller's stream below
            return e, y

        jobs = [lane["pool"].submit(encode, 2, lane["side"][0]), lane["pool"].submit(encode, 1, lane["side"][1])]
        # 1. durations (DurationPredictor + DurationProcessor.prediction_to_duration), then the frame offsets ON THE DEVICE:
        #    st / st4 are capacity layouts (host = upper bounds, device = the real cumulative frame counts)
        _, dur = eng.duration(sp, toks)
        st, st4, need = eng.frame_offsets(sp, dur, caps)
        (pe_enc, pe_style), (enc, style) = jobs[0].result(), jobs[1].result()
        main.wait_stream(lane["side"][0])
        main.wait_stream(lane["side"][1])
        # 2b. pitch / energy (PitchEnergyPredictor on the pe encoders' outputs)
        f0, en = eng.pitch_energy(sp, st, dur, pe_enc, pe_style)
        # 3b. speech predictor front (length regulator, x4 upsampling)
        asr = eng.length_regulate(sp, st4, dur, 4, enc, self.cfg.inter_dim)
        p4, e4 = eng.upsample4(st, st4, f0), eng.upsample4(st, st4, en)
        # 4. frame path.  Explicit noise is indexed by the REAL packed rows (as the outputs are): only its first rows are read.
        R = st4.rows
        if noise is None:
            noise = dict(prior_noise=torch.randn(R, 128, device=dev), src_noise=torch.randn(R * 75, device=dev),
                         init_phase=torch.rand(1, device=dev))
        if ev:
            ev[1].record(main)
        audio = eng.frame_path(st4, asr, p4, e4, style, noise["prior_noise"], noise["src_noise"], noise["init_phase"], batch_scope=False)
        if ev:
            ev[2].record(main)
        # the one host read of the call: the frame counts, together with the audio (the .cpu() waits for the stream)
        T = [int(v) for v in need.cpu().tolist()]
        if any(t > c for t, c in zip(T, caps)):  # truncated utterances: this call's output is invalid, and so is whatever the truncated
            # data left in the error word - it is read (which clears it) and dropped, then the call is repeated with what it asked for
            try:
                eng.check_status()
            except RuntimeError:
                pass
            e = CapacityOverflow(f"predicted frames {T} exceed the capacities {list(caps)}")
            e.need = T
            raise e
        eng.check_status()  # only a call whose capacities fitted reports the device-side status
        # adapt: capacity of the next calls = 1.25 x the largest frames-per-token ratio seen so far, rounded UP to a multiple of 2 (monotone
        # and coarse: it settles after the first calls, and equal inputs then get equal capacities, launch plans and results)
        if self._adapt:
            want = 1.25 * max(t / max(n, 1) for t, n in zip(T, L))
            self._ratio = min(float(self.MAX_FRAMES_PER_TOKEN), max(self._ratio, 2.0 * float(np.ceil(want / 2.0))))
        off = np.concatenate([[0], np.cumsum(T)]) * (4 * 75)
        waves = [audio[int(off[i]) : int(off[i + 1])] for i in range(len(L))]
        if return_details:
            Tm = int(sum(T))
            return waves, dict(durations=dur, frames=T, pitch=f0[:Tm], energy=en[:Tm], style=style, capacities=list(caps))
        return waves
