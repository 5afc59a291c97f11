"""The fused inference epilogue (mz_bias_act_nhwc) against plain PyTorch: the InferenceEngine with
fused=True must reproduce the nn.Module forward (network.py:146-163 contract)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_epilogue_kernel_vs_torch_fp32_reference():
    from muzero_amd import _lib as L
    import ctypes as C
    torch.manual_seed(0)
    for dt, code, tol in ((torch.float32, L.MZ_F32, 0.0), (torch.bfloat16, L.MZ_BF16, 1e-2), (torch.float16, L.MZ_F16, 2e-3)):
        for ch, n in ((128, 1000), (8, 4096), (16, 3)):
            y = torch.randn(n, ch, 8, 8, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
            z = torch.randn(n, ch, 8, 8, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
            b = torch.randn(ch, device="cuda").to(dt)
            for res in (None, z):
                for relu in (0, 1):
                    for bias in (None, b):
                        got = y.clone(memory_format=torch.preserve_format)
                        L.call("mz_bias_act_nhwc", C.c_void_p(got.data_ptr()), None if bias is None else C.c_void_p(bias.data_ptr()),
                               None if res is None else C.c_void_p(res.data_ptr()), got.numel(), ch, code, relu, L.current_stream())
                        ref = y.float() + (0 if bias is None else bias.float().view(1, -1, 1, 1)) + (0 if res is None else res.float())
                        ref = torch.relu(ref) if relu else ref
                        if dt == torch.float32:
                            assert torch.equal(got, ref)
                        else:
                            assert torch.equal(got, ref.to(dt))  # fp32 arithmetic, one round-to-nearest-even


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 6e-2)])
def test_fused_engine_matches_module_forward(dtype, tol):
    from muzero_amd.network import InferenceEngine, Network
    torch.manual_seed(1261)
    net = Network()
    B = 192
    fused = InferenceEngine(net, B, dtype, fused=True)
    plain = InferenceEngine(net, B, dtype, fused=False)
    img = (torch.rand(B, 9, 8, 8, device="cuda") < 0.3).float()
    x10 = torch.rand(B, 10, 8, 8, device="cuda").to(dtype)
    for a, b in ((fused.initial(img), plain.initial(img)), (fused.recurrent(x10), plain.recurrent(x10))):
        for u, v, name in zip(a, b, ("value", "policy", "hidden")):
            assert u.shape == v.shape and u.dtype == v.dtype
            assert (u.float() - v.float()).abs().max().item() <= tol * max(1.0, v.float().abs().max().item()), name
    # and against the reference-API forward of the fp32 module
    if dtype == torch.float32:
        with torch.no_grad():
            v, p, h = net.recurrent_inference(x10[:, :8], x10[:, 8:])
        a = fused.recurrent(x10)
        assert (a[0] - v.reshape(-1)).abs().max().item() < 1e-5 and (a[1] - p).abs().max().item() < 1e-5
