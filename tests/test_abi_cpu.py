"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/cerberus_hip.h declares; argument rejection paths that
never touch a GPU; the Python surface mirrors the reference's signatures."""
import ctypes
import inspect
import os
import re

import pytest
import torch

import cerberusnet_amd as ca
from cerberusnet_amd import _lib
from conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "cerberus_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cerberus_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.PROTOTYPES) == names  # binding covers the header exactly
    assert _lib.get().cerberus_abi_version() == _lib.ABI_VERSION == 7


def test_out_shape_matches_reference_rules():
    lib = _lib.get()
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    cases = {(16, 32, 4, 1, 4, 1, 1): (81, 16, 32), (16, 32, 4, 1, 10, 1, 1): (441, 4, 20),
             (64, 64, 3, 3, 20, 1, 2): (441, 28, 28), (17, 33, 4, 1, 4, 2, 2): (25, 9, 17)}
    for args, want in cases.items():
        assert lib.cerberus_correlation_out_shape(*args, ctypes.byref(oc), ctypes.byref(oh),
                                                  ctypes.byref(ow)) == 0
        assert (oc.value, oh.value, ow.value) == want
    # empty output -> CERB_EINVAL
    assert lib.cerberus_correlation_out_shape(4, 4, 0, 1, 4, 1, 1, ctypes.byref(oc),
                                              ctypes.byref(oh), ctypes.byref(ow)) == -1


def test_argument_rejection_without_gpu():
    lib = _lib.get()
    # bad dtype, stride1 != 1 in backward, null pointers: all rejected before any launch
    assert lib.cerberus_correlation_forward(None, None, None, 1, 4, 8, 8, 4, 1, 4, 1, 1, 1, 9,
                                            None) == -2
    assert lib.cerberus_correlation_backward(None, None, None, None, None, 1, 4, 8, 8, 4, 1, 4,
                                             2, 1, 1, 0, None) == -3
    assert lib.cerberus_correlation_forward(None, None, None, 1, 4, 8, 8, 4, 1, 4, 1, 1, 1, 0,
                                            None) == -1
    assert lib.cerberus_flow_warp_forward(None, None, None, 1, 4, 8, 8, 7, 0, 0, None) == -4
    assert lib.cerberus_flow_warp_forward(None, None, None, 0, 4, 8, 8, 1, 0, 0, None) == 0
    assert lib.cerberus_flow_warp_context_bytes(2, 8, 16) == 2 * 4 * 1 * 16 + 2 * 2 * 8 * 16 * 4
    assert lib.cerberus_flow_warp_backward_workspace_bytes(2, 6, 8, 16) == 16 + (2 * 4 * 1 * 16 + 2 * 2 * 8 * 16 * 4)
    assert b"stride1" in lib.cerberus_error_string(-3)
    assert lib.cerberus_set_option(b"no_such_key", 1) == -1


def test_python_surface_mirrors_reference_signatures():
    # correlation.py:61-62
    sig = inspect.signature(ca.Correlation.__init__)
    assert list(sig.parameters)[1:] == ["pad_size", "kernel_size", "max_displacement",
                                        "stride1", "stride2", "corr_multiply"]
    assert [p.default for p in list(sig.parameters.values())[1:]] == [0, 0, 0, 1, 2, 1]
    # correlation.py:31-32
    fsig = inspect.signature(ca.CorrelationFunction.forward)
    assert [p.default for p in list(fsig.parameters.values())[3:]] == [3, 3, 20, 1, 2, 1]
    # UnFlowLoss.py:83
    wsig = inspect.signature(ca.flow_warp)
    assert list(wsig.parameters) == ["image", "flow12", "pad", "mode"]
    assert wsig.parameters["pad"].default == "border"
    assert wsig.parameters["mode"].default == "bilinear"
    m = ca.Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1,
                       corr_multiply=1)
    assert len(m.state_dict()) == 0 and len(list(m.parameters())) == 0


def test_op_schemas_match_reference_registration():
    # correlation_cuda.cpp:3-4, 28-30, 45-48
    s = str(torch.ops.cerberus.correlation.default._schema)
    assert s.startswith("cerberus::correlation(Tensor input1, Tensor input2, int pad_size, "
                        "int kernel_size, int max_displacement, int stride1, int stride2, "
                        "int corr_type_multiply) -> Tensor")
    b = str(torch.ops.cerberus.correlation_backward.default._schema)
    assert "Tensor gradOutput" in b and b.endswith("-> Tensor[]")


def test_cpu_tensors_fail_loudly():
    x = torch.randn(1, 4, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        torch.ops.cerberus.correlation(x, x, 4, 1, 4, 1, 1, 1)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        ca.flow_warp(x, torch.zeros(1, 2, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        ca.Correlation(4, 1, 4, 1, 1, 1)(x, x)


def test_meta_shapes():
    x = torch.empty(2, 8, 16, 32, device="meta")
    assert torch.ops.cerberus.correlation(x, x, 4, 1, 4, 1, 1, 1).shape == (2, 81, 16, 32)
    assert torch.ops.cerberus.correlation(x, x, 4, 1, 10, 1, 1, 1).shape == (2, 441, 4, 20)
    g1, g2 = torch.ops.cerberus.correlation_backward(
        x, x, torch.empty(2, 81, 16, 32, device="meta"), 4, 1, 4, 1, 1, 1)
    assert g1.shape == x.shape and g2.shape == x.shape


def test_warp_context_ops_are_registered_and_fail_loudly_on_cpu():
    """The training-path warp ops (forward that saves its backward context, backward that
    consumes it): schemas, meta shapes consistent with the C ABI's size function, and no
    silent CPU path -- with and without autograd."""
    lib = _lib.get()
    s = str(torch.ops.cerberus.flow_warp_ctx.default._schema)
    assert s.endswith("-> (Tensor, Tensor)")
    b = str(torch.ops.cerberus.flow_warp_backward_ctx.default._schema)
    assert "Tensor context" in b and b.endswith("-> Tensor[]")
    x = torch.empty(2, 8, 16, 32, device="meta")
    f = torch.empty(2, 2, 16, 32, device="meta")
    out, ctx = torch.ops.cerberus.flow_warp_ctx(x, f, 1, 0)
    assert out.shape == x.shape
    assert ctx.numel() * ctx.element_size() >= lib.cerberus_flow_warp_context_bytes(2, 16, 32)
    gi, gf = torch.ops.cerberus.flow_warp_backward_ctx(x, f, ctx, x, 1, 0, True, True)
    assert gi.shape == x.shape and gf.shape == f.shape
    xc = torch.randn(1, 4, 8, 8, requires_grad=True)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        ca.flow_warp(xc, torch.zeros(1, 2, 8, 8))            # autograd path -> flow_warp_ctx
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        torch.ops.cerberus.flow_warp_backward_ctx(xc.detach(), torch.zeros(1, 2, 8, 8),
                                                  torch.zeros(16, dtype=torch.int64),
                                                  xc.detach(), 1, 0, True, True)
    # context / workspace sizes grow with the shape and never go negative
    assert lib.cerberus_flow_warp_context_bytes(0, 8, 8) == 0
    assert lib.cerberus_flow_warp_context_bytes(-1, 8, 8) == 0
    # the meta function's pure-Python size (fake-tensor tracing must not need the .so) equals the library's
    from cerberusnet_amd.ops import _warp_context_bytes
    for shape in ((1, 1, 1), (2, 8, 16), (3, 7, 33), (4, 128, 256), (4, 512, 1024), (1, 1025, 31), (0, 8, 8), (-1, 8, 8)):
        assert _warp_context_bytes(*shape) == lib.cerberus_flow_warp_context_bytes(*shape), shape
    assert (lib.cerberus_flow_warp_backward_workspace_bytes(4, 32, 128, 256)
            == 16 + lib.cerberus_flow_warp_context_bytes(4, 128, 256))


def test_correlation_torch_equals_oracle_restatement():
    import oracle
    a, b = torch.randn(2, 5, 6, 7), torch.randn(2, 5, 6, 7)
    assert torch.equal(ca.CorrelationTorch(2)(a, b), oracle.correlation_torch_ref(a, b, 2))


def test_product_never_imports_oracle():
    """The shipped package must not reach into oracle/ (checked textually)."""
    pkg = os.path.join(REPO, "cerberusnet_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "corr_oracle" not in text, f


def test_the_product_library_holds_dispatched_code_only():
    """the measured-and-rejected kernels (the persistent pipelined forward, the column-walking / two-item
    backward variants) are compiled into lib/libcerberus_hip_experiments.so only; both libraries export the same C ABI."""
    exp_path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libcerberus_hip_experiments.so")
    assert os.path.exists(exp_path), "python -m cerberusnet_amd.build --experiments (or __graft_entry__.build())"
    v = ctypes.c_int(-1)
    assert _lib.get().cerberus_get_option(b"experiments_build", ctypes.byref(v)) == 0 and v.value == (
        1 if "experiments" in os.path.basename(_lib.LIB_PATH) else 0)
    exp = ctypes.CDLL(exp_path)
    assert exp.cerberus_get_option(b"experiments_build", ctypes.byref(v)) == 0 and v.value == 1
    for name in _declared_symbols():
        assert hasattr(exp, name), name
    prod_bytes, exp_bytes = open(os.path.join(os.path.dirname(exp_path), "libcerberus_hip.so"), "rb").read(), open(exp_path, "rb").read()
    for marker in (b"corr_fwd_d4_pipe", b"corr_bwd_d4_col_4x64", b"corr_bwd_d4_strip_w256_2items"):
        assert marker in exp_bytes and marker not in prod_bytes, marker
    from cerberusnet_amd import build
    assert "corr_fwd_pipe.hip" not in build.SOURCES and "corr_fwd_pipe.hip" in build.EXPERIMENT_SOURCES
