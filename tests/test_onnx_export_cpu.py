"""SURVEY 8(f)-4, export half, on CPU: the symbolics of cerberusnet_amd.utilities.onnx_export emit the
reference's custom node types (nnet_training/utilities/onnx_export.py:18-28).  The HIP ops have no CPU
path, so the symbolics are exercised through stand-in ops of the same schemas in a scratch namespace;
tests/test_pwchead_gpu.py exports the real head."""
import warnings

import torch

from cerberusnet_amd.utilities import onnx_export as oe

_lib = torch.library.Library("cerb_export_test", "DEF")
_lib.define("correlation(Tensor input1, Tensor input2, int pad_size, int kernel_size, int max_displacement, "
            "int stride1, int stride2, int corr_type_multiply) -> Tensor")
_lib.impl("correlation", lambda a, b, *r: torch.cat([a[:, :1] * b[:, :1]] * 81, 1), "CompositeExplicitAutograd")
_lib.define("flow_warp(Tensor image, Tensor flow, int pad_mode, int interp_mode) -> Tensor")
_lib.impl("flow_warp", lambda i, f, p, m: i * 1.0, "CompositeExplicitAutograd")


class _Level(torch.nn.Module):
    def forward(self, f1, f2, flow):
        warped = torch.ops.cerb_export_test.flow_warp(f2, flow, 1, 0)
        vol = torch.ops.cerb_export_test.correlation(f1, warped, 4, 1, 4, 1, 1, 1)
        return torch.nn.functional.leaky_relu(vol, 0.1)


def test_symbolics_emit_the_reference_custom_nodes(tmp_path):
    torch.onnx.register_custom_op_symbolic("cerb_export_test::correlation", oe.correlation_op, oe.OPSET)
    torch.onnx.register_custom_op_symbolic("cerb_export_test::flow_warp", oe.flow_warp_op, oe.OPSET)
    f = torch.randn(1, 4, 6, 8)
    path = str(tmp_path / "level.onnx")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with oe._without_onnxscript_splice():
            torch.onnx.export(_Level(), (f, f, torch.randn(1, 2, 6, 8)), path, opset_version=oe.OPSET,
                              dynamo=False, do_constant_folding=False)
    blob = open(path, "rb").read()
    assert b"cerberus" in blob and b"correlation" in blob and b"grid_sampler" in blob
    for attr in (b"pad_size", b"kernel_size", b"max_displacement", b"stride1", b"stride2", b"corr_multiply",
                 b"interpolation_mode", b"padding_mode", b"align_corners", b"LeakyRelu"):
        assert attr in blob, attr
    # the stand-in namespace must not leak into the graph: only the reference's node domains do
    assert b"cerb_export_test" not in blob
