"""ONNX export of a model that uses the hot-path ops, with the reference's custom nodes
(reference: ``nnet_training/utilities/onnx_export.py:18-28,45-46``).

The reference exports its networks with two custom symbolics -- ``cerberus::correlation`` (attributes
``pad_size_i, kernel_size_i, max_displacement_i, stride1_i, stride2_i, corr_multiply_i``) and
``torch::grid_sampler`` (``interpolation_mode_i, padding_mode_i, align_corners_i``) -- which its TensorRT
runtime resolves with plugins (``runtime/cerberus_net/trt_plugins/correlation.cu``, ``grid_sampler.cu``).
``register_symbolics`` registers the same two node types for THIS package's ops, so that a graph exported
here carries exactly the reference's node names and attributes:

  * ``cerberus::correlation``            -> one ``cerberus::correlation`` node (the drop-in op);
  * ``cerberus::correlation_leaky``      -> that node followed by ``LeakyRelu`` (the fused epilogue unfused);
  * ``cerberus::flow_warp``              -> what tracing the reference's ``flow_warp`` yields
    (``UnFlowLoss.py:83-94``): a constant pixel mesh + flow, normalised by (W-1, H-1), fed to a
    ``torch::grid_sampler`` node with ``align_corners_i = 0``.

Export the head with ``fuse_concat=False, fuse_upsample=False`` (the reference's op sequence: ``torch.cat``
and ``F.interpolate`` are stock ops the exporter knows).  On MI355X itself the low-latency inference path
is ``cerberusnet_amd.graphs.GraphedFlowInference`` (a hipGraph of the PyTorch-ROCm forward on the HIP
ops); an ONNX runtime for these nodes (MIGraphX custom ops) is not part of this repository: neither
MIGraphX nor onnx / onnxruntime ship in the build image, so nothing here can be run against one.
"""
import torch
from torch.onnx import symbolic_helper as sym_help

OPSET = 11   # onnx_export.py:62


@sym_help.parse_args("v", "v", "i", "i", "i", "i", "i", "i")
def correlation_op(g, input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2,
                   corr_multiply):
    """Same node and attribute names as the reference's ``correlation_op`` (onnx_export.py:18-23)."""
    return g.op("cerberus::correlation", input1, input2, pad_size_i=pad_size, kernel_size_i=kernel_size,
                max_displacement_i=max_displacement, stride1_i=stride1, stride2_i=stride2,
                corr_multiply_i=corr_multiply)


@sym_help.parse_args("v", "v", "i", "i", "i", "i", "i", "i", "f")
def correlation_leaky_op(g, input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2,
                         corr_multiply, negative_slope):
    out = g.op("cerberus::correlation", input1, input2, pad_size_i=pad_size, kernel_size_i=kernel_size,
               max_displacement_i=max_displacement, stride1_i=stride1, stride2_i=stride2,
               corr_multiply_i=corr_multiply)
    return g.op("LeakyRelu", out, alpha_f=negative_slope)


@sym_help.parse_args("v", "v", "i", "i")
def flow_warp_op(g, image, flow, pad_mode, interp_mode):
    """flow_warp(image, flow) as the reference's graph has it: grid = 2 (mesh + flow) / (W-1, H-1) - 1,
    ``torch::grid_sampler(image, grid, mode, padding, align_corners=False)`` (UnFlowLoss.py:11-32,83-94,
    onnx_export.py:25-28).  ATen's enum values: interpolation 0 bilinear / 1 nearest, padding 0 zeros /
    1 border / 2 reflection -- the same codes as cerb_interp_mode / cerb_pad_mode."""
    # H, W from whichever operand carries static sizes in the trace (the image is a graph input; the flow
    # comes out of nodes behind a custom op, whose shapes the exporter does not infer)
    sizes = None
    for t in (flow, image):
        sz = sym_help._get_tensor_sizes(t)
        if sz is not None and len(sz) == 4 and sz[2] is not None and sz[3] is not None:
            sizes = sz
            break
    if sizes is None:
        raise RuntimeError("cerberus::flow_warp needs static H, W to export (trace with concrete inputs)")
    h, w = int(sizes[2]), int(sizes[3])
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32),
                            indexing="ij")
    mesh = g.op("Constant", value_t=torch.stack([xs, ys], 0).unsqueeze(0))             # (1, 2, H, W), ch0 = x
    scale = g.op("Constant", value_t=torch.tensor([2.0 / max(w - 1, 1), 2.0 / max(h - 1, 1)],
                                                  dtype=torch.float32).view(1, 2, 1, 1))
    one = g.op("Constant", value_t=torch.tensor(1.0))
    v = g.op("Add", mesh, flow)
    grid = g.op("Sub", g.op("Mul", v, scale), one)
    grid = g.op("Transpose", grid, perm_i=[0, 2, 3, 1])                                  # (B, H, W, 2)
    return g.op("torch::grid_sampler", image, grid, interpolation_mode_i=interp_mode,
                padding_mode_i=pad_mode, align_corners_i=0)


def register_symbolics(opset: int = OPSET) -> None:
    """Call once before ``torch.onnx.export(..., opset_version=opset, dynamo=False)``."""
    import cerberusnet_amd  # noqa: F401  (registers torch.ops.cerberus.*)
    torch.onnx.register_custom_op_symbolic("cerberus::correlation", correlation_op, opset)
    torch.onnx.register_custom_op_symbolic("cerberus::correlation_leaky", correlation_leaky_op, opset)
    torch.onnx.register_custom_op_symbolic("cerberus::flow_warp", flow_warp_op, opset)


class _HeadForExport(torch.nn.Module):
    """PWCNetHead takes ``(concat, [features])`` pairs; the exporter wants flat tensor inputs."""

    def __init__(self, head, levels):
        super().__init__()
        self.head, self.levels = head, levels

    def forward(self, *feats):
        p1, p2 = list(feats[:self.levels]), list(feats[self.levels:])
        return tuple(self.head((None, p1), (None, p2)))


def export_flow_head(head: torch.nn.Module, pyr1, pyr2, path: str, opset: int = OPSET) -> str:
    """Export ``head(pyr1, pyr2)`` (eval mode, forward only) to ``path``.  The head must run the
    reference's op sequence (``fuse_concat=False, fuse_upsample=False``); the tensors must live where
    the ops can run (an MI355X: there is no CPU path)."""
    if getattr(head, "fuse_concat", False) or getattr(head, "fuse_upsample", False):
        raise ValueError("export the head with fuse_concat=False, fuse_upsample=False (reference op sequence)")
    register_symbolics(opset)
    wrapper = _HeadForExport(head.eval(), len(pyr1))
    with torch.no_grad(), _without_onnxscript_splice():
        torch.onnx.export(wrapper, tuple(pyr1) + tuple(pyr2), path, opset_version=opset, dynamo=False,
                          do_constant_folding=False)
    return path


class _without_onnxscript_splice:
    """torch's TorchScript exporter serialises the graph itself; its last step, splicing onnxscript
    functions into the proto, imports the ``onnx`` package although a graph without such functions
    (ours) passes through unchanged.  Where ``onnx`` is missing (this image) that step is skipped."""

    def __enter__(self):
        self.mod = self.orig = None
        try:
            import onnx  # noqa: F401
        except ImportError:
            try:
                from torch.onnx._internal.torchscript_exporter import onnx_proto_utils as mod
            except ImportError:        # older torch: torch.onnx._onnx_proto_utils? leave the export to fail loudly
                return self
            self.mod, self.orig = mod, mod._add_onnxscript_fn
            mod._add_onnxscript_fn = lambda proto, custom_opsets: proto
        return self

    def __exit__(self, *exc):
        if self.mod is not None:
            self.mod._add_onnxscript_fn = self.orig
        return False
