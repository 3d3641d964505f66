"""Tuning aid: TF/s of conv_gemm_f32 per layer shape and tile (runs on the GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylish_tts_amd import _lib
lib = C.CDLL(os.environ.get("STTS_LIB", _lib.LIB_PATH))  # STTS_LIB: e.g. a -DSTTS_GEMM_TRACE build (ablation bits of TUNE)
lib.stts_bench_gemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
torch.zeros(1).cuda()
shapes = [  # name, cin, cout, k
    ("out_conv 768->1025 k7", 768, 1025, 7), ("out_conv 768->1024 k7", 768, 1024, 7),
    ("pwconv1 512->1536", 512, 1536, 1), ("pwconv2 1536->512", 1536, 512, 1),
    ("dec conv1 578->512 k3", 578, 512, 3), ("dec conv2 512->512 k3", 512, 512, 3),
    ("prior 1025->256 k7", 1025, 256, 7), ("flow in 128->256 k5", 128, 256, 5), ("flow rs 128->256", 128, 256, 1),
    ("square 4096 (1 utt)", 4096, 4096, 1),
    ("wino plane 768->1024 k1", 768, 1024, 1), ("wino plane 608->512 k1", 608, 512, 1),  # B=20 / B=12 x 960 rows = the component planes at B=8
    ("small asr_res 128->64", 128, 64, 1), ("small prior 512->256", 512, 256, 1), ("small pre 64->128", 64, 128, 1), ("small post_flow 128->512", 128, 512, 1),
]
B, T4 = int(os.environ.get("B", 8)), int(os.environ.get("T4", 960))
flt = os.environ.get("SHAPES")
tiles = [1000 + int(t[1:]) if t.startswith("x") else 2000 + int(t[1:]) if t.startswith("p") else int(t) for t in os.environ.get("TILES", "2,5,6").split(",")]
for name, cin, cout, k in shapes:
    if flt and not any(f in name for f in flt.split(",")):
        continue
    line = f"{name:26s}"
    for tile in tiles:
        ms = C.c_double()
        nu, rows = (1, 4096) if name.startswith("square") else (B, T4)
        # tile "x5" = tile 5 of the split-fp32 contraction (TUNE bit 11); a plain number = the f32 matrix cores (100 + t: whatever STTS_NO_X3 says)
        tune = int(os.environ.get('TUNE', 0))
        # "p25" .. "p28": split-fp32 tiles that read PRE-SPLIT activation planes (TUNE bits 11 + 12)
        t = tile % 1000
        rc = lib.stts_bench_gemm(None, nu, rows, cin, cout, k, t, 10, C.byref(ms), tune | (2048 if tile >= 1000 else 0) | (4096 if tile >= 2000 else 0))
        if rc != 0:  # this tile does not apply to the shape (e.g. 256-column tiles need cout padded to 256)
            line += f"  tile{tile}:      n/a            "
            continue
        fl = 2.0 * nu * rows * cout * cin * k
        line += f"  {'p' + str(tile - 2000) if tile >= 2000 else 'x' + str(tile - 1000) if tile >= 1000 else 't' + str(tile)}: {ms.value*1e3:7.1f} us {fl/ms.value/1e9:6.1f} TF"
    print(line, flush=True)
