#!/usr/bin/env python3
"""Build the library's HOST side with AddressSanitizer + UBSan against the HIP stub and run the sanitizer driver.

    python tests/asan/build_and_run.py [build_dir]

Steps: (1) hipcc --cuda-host-only -fsanitize=address,undefined -c for every translation unit of the library (no device code is generated or loaded),
(2) link it with hip_stub.cpp into libstylish_hip_asan.so (the fat-binary symbol the host code references is defined as an
empty blob), (3) dump the synthetic weights of every inference module to a binary file, (4) run asan_driver on it.
Exit code 0 = no sanitizer report and every stage accepted its workspace for every shape."""
import os
import struct
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main(build_dir):
    os.makedirs(build_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-O1", "-g", "-std=c++17", "-fPIC"]
    # every translation unit of the library (__graft_entry__.UNITS: api.hip + one unit per operand form of the contraction kernels), host side only, in parallel
    import __graft_entry__ as entry

    objs, procs = [], []
    for u in entry.UNITS:
        o = os.path.join(build_dir, u.replace(".hip", "_host.o"))
        objs.append(o)
        procs.append(subprocess.Popen([hipcc, "--offload-arch=gfx950", "--cuda-host-only", *san, "-c", os.path.join(ROOT, "stylish_tts_amd", "csrc", u), "-o", o]))
    if any(p.wait() != 0 for p in procs):
        raise subprocess.CalledProcessError(1, "hipcc --cuda-host-only")
    fatbin = sorted({s for o in objs for s in subprocess.check_output(["nm", "-u", o], text=True).split() if s.startswith("__hip_fatbin")})
    cxx = os.environ.get("CXX_ASAN", "/opt/rocm/lib/llvm/bin/clang++")  # the same clang, as a plain C++ compiler / linker driver
    stub = os.path.join(build_dir, "hip_stub.o")
    subprocess.check_call([cxx, *san, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c", os.path.join(HERE, "hip_stub.cpp"), "-o", stub])
    lib = os.path.join(build_dir, "libstylish_hip_asan.so")
    defsym = [f"-Wl,--defsym={s}=stts_stub_fatbin" for s in fatbin]
    subprocess.check_call([cxx, "-shared", *san, *objs, stub, *defsym, "-o", lib])
    exe = os.path.join(build_dir, "asan_driver")
    subprocess.check_call([cxx, *san, os.path.join(HERE, "asan_driver.cpp"), lib, f"-Wl,-rpath,{build_dir}", "-o", exe])

    # weights: every inference module's synthetic state dict, in the library's naming (module + "." + key)
    from stylish_tts_amd import _lib, params
    from stylish_tts_amd.config import load_model_config

    cfg = load_model_config()
    wpath = os.path.join(build_dir, "weights.bin")
    with open(wpath, "wb") as f:
        f.write(bytes(_lib.dims_from_config(cfg)))
        for m in params.MODULE_SPECS:
            sd = params.synth_state_dict(params.module_spec(m, cfg), 0, prefix=m + ".")
            for k, v in sd.items():
                name = (m + "." + k).encode()
                shape = v.shape if v.ndim else (1,)
                f.write(struct.pack("<i", len(name)) + name + struct.pack("<i", len(shape)) + struct.pack(f"<{len(shape)}q", *shape))
                f.write(v.astype("<f4").tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, wpath], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print(r.stdout[-6000:])
    return r.returncode


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/stts_asan_build"))
