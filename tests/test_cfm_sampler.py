"""CfmSampler (Euler ODE solver of the flow-matching denoiser, models/cfm/cfm.py:44-84) against vectors produced by the
reference sampler with a closed-form estimator (tests/golden/gen_golden.py:cfm_golden)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import stylish_oracle as O

STEPS = (1, 4, 7, 32)


def _estimator_np(cond, gain):
    def f(x, t):
        return ((cond - x) * (np.float32(0.5) + t.reshape(-1, 1, 1)) * np.float32(gain) + np.sin(np.float32(3.0) * x)).astype(np.float32)

    return f


def test_oracle_time_grid_is_torch_linspace():
    torch = pytest.importorskip("torch")
    # (torch's CPU kernel forms long grids per SIMD vector as base + lane * step, so only short grids are compared
    #  here; its GPU kernel, which is what the sampler runs, uses the per-element formula below)
    for n in (1, 2, 3, 4, 7, 10, 32, 33):
        step = np.float32(1.0) / np.float32(n)
        idx = np.arange(n + 1)
        grid = np.where(idx < (n + 1) // 2, idx.astype(np.float32) * step, np.float32(1.0) - (n - idx).astype(np.float32) * step).astype(np.float32)
        assert np.array_equal(grid, torch.linspace(0, 1, n + 1).numpy()), n


@pytest.mark.parametrize("n", STEPS)
def test_oracle_matches_reference_sampler(n):
    g = load_golden("cfm_euler")
    y = O.cfm_solve_euler(g[f"z{n}"], n, _estimator_np(g[f"cond{n}"], 1.7), temperature=0.8)
    # numpy's and torch's sin differ by an ulp here and there; the update arithmetic itself is exact
    assert np.abs(y - g[f"y{n}"]).max() < 2e-5 * max(1.0, np.abs(g[f"y{n}"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("n", STEPS)
def test_hip_sampler_matches_reference(n):
    torch = pytest.importorskip("torch")
    from stylish_tts_amd.euler_sampler import CfmSampler

    g = load_golden("cfm_euler")
    z, cond = torch.from_numpy(g[f"z{n}"]).cuda(), torch.from_numpy(g[f"cond{n}"]).cuda()
    calls = []

    def estimator(x, t, mask, cond, gain):
        calls.append(float(t[0]))
        assert mask is None and t.shape == (x.shape[0],)
        return (cond - x) * (0.5 + t.reshape(-1, 1, 1)) * gain + torch.sin(3.0 * x)

    y = CfmSampler(estimator)(z, None, n, temperature=0.8, cond=cond, gain=1.7)
    ref = g[f"y{n}"]
    assert np.abs(y.cpu().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    assert len(calls) == n and calls[0] == 0.0
    assert np.allclose(calls, torch.linspace(0, 1, n + 1)[:-1].numpy(), atol=1e-6)


@pytest.mark.gpu
def test_euler_step_is_mul_then_add():
    """x += dt * v with two roundings, bit-identical to torch's `x + dt * v`."""
    import ctypes as C

    torch = pytest.importorskip("torch")
    from stylish_tts_amd import _lib

    lib = _lib.load()
    gen = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(100003, generator=gen).cuda()
    v = torch.randn(100003, generator=gen).cuda()
    dt = 0.14285715
    want = x + torch.tensor(dt, dtype=torch.float32, device="cuda") * v
    _lib.check(lib.stts_euler_step(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(x.data_ptr()), C.c_void_p(v.data_ptr()), C.c_float(dt),
                                   x.numel()))
    assert torch.equal(x, want)
