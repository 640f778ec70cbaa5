// cerberus_runtime.hpp -- torch-free MI355X inference host for the two custom ONNX nodes of an exported
// Cerberus model: `cerberus::correlation` and the flow warp that the
// reference exports as `torch::grid_sampler` behind its mesh / norm_grid ops
// (/root/reference/nnet_training/utilities/onnx_export.py:18-28).
//
// Counterpart of the reference's TensorRT plugins:
//   CorrelationLayer  <->  runtime/cerberus_net/trt_plugins/correlation.{hpp,cpp,cu}
//                          (IPluginV2DynamicExt: getOutputDimensions correlation.cpp, enqueue correlation.cu:94-166)
//   FlowWarpLayer     <->  runtime/cerberus_net/trt_plugins/grid_sampler.cu:238-271 (enqueue)
// with the same plugin-style surface (output dims from input dims, workspace size, enqueue on a stream,
// fp32 / fp16) -- minus everything TensorRT: the layers call the C ABI of libcerberus_hip.so
// (include/cerberus_hip.h) on a HIP stream.  Differences that are the point of the rewrite:
//   * enqueue never synchronises the stream (correlation.cu:105,124-125 synchronise three times per call),
//     needs no workspace (the plugin keeps two zero-padded NHWC copies, correlation.cu:100-118) and is
//     therefore capturable: FlowPyramidGraph records warp -> correlation (+ LeakyReLU) for all pyramid
//     levels once and replays them as ONE hipGraph per frame pair;
//   * the warp takes the FLOW (pixels), not a normalised grid: mesh + norm_grid + unnormalise are fused
//     into the kernel with the training-time semantics (UnFlowLoss.py:83-94, quirk Q2) -- the TRT plugin
//     computes align_corners=True coordinates there (grid_sampler.cu:55-58), i.e. deployment disagreed with
//     training by up to half a pixel at the borders; this runtime does not.
// Header-only; link with -lcerberus_hip and the HIP runtime (hipcc).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cerberus_hip.h"

namespace cerberus_rt {

struct Dims4 { int n = 0, c = 0, h = 0, w = 0; int64_t count() const { return int64_t(n) * c * h * w; } };

inline size_t dtype_bytes(int dtype) {
    switch (dtype) {
        case CERB_F32: return 4;
        case CERB_F16: case CERB_BF16: return 2;
        case CERB_F64: return 8;
    }
    throw std::invalid_argument("cerberus_rt: unknown dtype");
}

inline void check(int code, const char *what) {
    if (code != 0) throw std::runtime_error(std::string(what) + ": " + cerberus_error_string(code));
}
inline void check_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// cerberus::correlation(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2,
// corr_type_multiply) -- the attribute set of the exported node (onnx_export.py:18-23) and of the TRT
// plugin's field collection (correlation.cpp: PluginFieldCollection).
class CorrelationLayer {
  public:
    CorrelationLayer(int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                     int corr_type_multiply = 1, float leaky_slope = 1.0f)
        : pad_(pad_size), k_(kernel_size), d_(max_displacement), s1_(stride1), s2_(stride2),
          mult_(corr_type_multiply), slope_(leaky_slope) {}
    int getNbOutputs() const { return 1; }
    // correlation_cuda.cpp:6-14 (the plugin's getOutputDimensions restates it with IExprBuilder)
    Dims4 getOutputDimensions(const Dims4 &in) const {
        int oc = 0, oh = 0, ow = 0;
        check(cerberus_correlation_out_shape(in.h, in.w, pad_, k_, d_, s1_, s2_, &oc, &oh, &ow),
              "correlation output shape");
        return Dims4{in.n, oc, oh, ow};
    }
    size_t getWorkspaceSize(const Dims4 &) const { return 0; }
    bool supportsFormat(int dtype) const { return dtype == CERB_F32 || dtype == CERB_F16 || dtype == CERB_BF16; }
    // inputs[0], inputs[1]: (N, C, H, W) device pointers; outputs[0]: (N, D*D, oH, oW).  Asynchronous.
    int enqueue(const Dims4 &in, int dtype, const void *const *inputs, void *const *outputs, void * /*workspace*/,
                hipStream_t stream) const {
        return cerberus_correlation_forward_ex(inputs[0], inputs[1], outputs[0], in.n, in.c, in.h, in.w, pad_, k_,
                                               d_, s1_, s2_, slope_, 0, dtype, stream);
    }
    int corrTypeMultiply() const { return mult_; }   // accepted and ignored, as in the reference (.cu:244-324)
  private:
    int pad_, k_, d_, s1_, s2_, mult_;
    float slope_;   // 1.0: plain correlation; 0.1: the LeakyReLU of pwcnet_sfd.py:182 fused into the store
};

// flow_warp(image, flow, pad, mode) (UnFlowLoss.py:83-94): what the exported graph spells as
// mesh + flow -> norm_grid -> torch::grid_sampler(input, grid, interp, padding, align_corners=False).
class FlowWarpLayer {
  public:
    explicit FlowWarpLayer(int pad_mode = CERB_PAD_BORDER, int interp_mode = CERB_INTERP_BILINEAR)
        : pad_(pad_mode), interp_(interp_mode) {}
    int getNbOutputs() const { return 1; }
    Dims4 getOutputDimensions(const Dims4 &image) const { return image; }
    size_t getWorkspaceSize(const Dims4 &) const { return 0; }
    bool supportsFormat(int dtype) const { return dtype == CERB_F32 || dtype == CERB_F16 || dtype == CERB_BF16; }
    // inputs[0]: image (N, C, H, W); inputs[1]: flow (N, 2, H, W) in pixels, channel 0 = x, of `flow_dtype`
    int enqueue(const Dims4 &image, int dtype, int flow_dtype, const void *const *inputs, void *const *outputs,
                void * /*workspace*/, hipStream_t stream) const {
        return cerberus_flow_warp_forward_ctx(inputs[0], inputs[1], outputs[0], nullptr, 0, image.n, image.c, image.h,
                                              image.w, pad_, interp_, dtype, flow_dtype, stream);
    }
  private:
    int pad_, interp_;
};

// One pyramid level of PWCNetHead's inference path (pwcnet_sfd.py:176-182): warped = flow_warp(f2, flow);
// cost = leaky_relu(correlation(f1, warped), 0.1).  Level 0 has no flow (f2 is correlated as it is).
struct PyramidLevel {
    Dims4 feat;             // (N, C, H, W) of f1 / f2
    bool has_flow = true;
    void *f1 = nullptr, *f2 = nullptr, *flow = nullptr, *warped = nullptr, *cost = nullptr;   // device
    Dims4 cost_dims;
};

// All levels, both ops, captured once into a hipGraph and replayed per frame pair: the analogue of the
// TensorRT engine's enqueueV2 over these plugin layers (cerberus.cpp:317-323), for the part of the network
// this package owns.  Owns its device buffers.
class FlowPyramidGraph {
  public:
    FlowPyramidGraph(const std::vector<Dims4> &levels, int dtype, const CorrelationLayer &corr,
                     const FlowWarpLayer &warp)
        : dtype_(dtype), corr_(corr), warp_(warp) {
        // a throw from a constructor skips the destructor: release what was acquired so far ourselves
        try {
            const size_t e = dtype_bytes(dtype);
            for (const Dims4 &d : levels)
                if (d.n <= 0 || d.c <= 0 || d.h <= 0 || d.w <= 0) throw std::runtime_error("FlowPyramidGraph: level dimensions must be positive");
            check_hip(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
            levels_.reserve(levels.size());
            for (size_t l = 0; l < levels.size(); ++l) {
                levels_.emplace_back();          // registered first: a failing hipMalloc below leaves its siblings to release()
                PyramidLevel &lv = levels_.back();
                lv.feat = levels[l];
                lv.has_flow = l > 0;
                lv.cost_dims = corr_.getOutputDimensions(lv.feat);
                const size_t fbytes = size_t(lv.feat.count()) * e;
                check_hip(hipMalloc(&lv.f1, fbytes), "hipMalloc");
                check_hip(hipMalloc(&lv.f2, fbytes), "hipMalloc");
                check_hip(hipMalloc(&lv.cost, size_t(lv.cost_dims.count()) * e), "hipMalloc");
                if (lv.has_flow) {
                    check_hip(hipMalloc(&lv.flow, size_t(lv.feat.n) * 2 * lv.feat.h * lv.feat.w * e), "hipMalloc");
                    check_hip(hipMalloc(&lv.warped, fbytes), "hipMalloc");
                }
            }
        } catch (...) {
            release();
            throw;
        }
    }
    FlowPyramidGraph(const FlowPyramidGraph &) = delete;
    FlowPyramidGraph &operator=(const FlowPyramidGraph &) = delete;
    ~FlowPyramidGraph() { release(); }
    std::vector<PyramidLevel> &levels() { return levels_; }
    hipStream_t stream() const { return stream_; }
    int dtype() const { return dtype_; }

    // the op sequence on `s` (eagerly, or inside a capture)
    void enqueue(hipStream_t s) const {
        for (const auto &lv : levels_) {
            const void *second = lv.f2;
            if (lv.has_flow) {
                const void *win[2] = {lv.f2, lv.flow};
                void *wout[1] = {lv.warped};
                check(warp_.enqueue(lv.feat, dtype_, dtype_, win, wout, nullptr, s), "flow_warp enqueue");
                second = lv.warped;
            }
            const void *cin[2] = {lv.f1, second};
            void *cout_[1] = {lv.cost};
            check(corr_.enqueue(lv.feat, dtype_, cin, cout_, nullptr, s), "correlation enqueue");
        }
    }
    // record once ... (a second call re-records: the old graph and its executable are destroyed first)
    void capture() {
        drop_graph();
        enqueue(stream_);                                   // warm-up (lazy module load is not capturable)
        check_hip(hipStreamSynchronize(stream_), "warm-up");
        check_hip(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal), "begin capture");
        try {
            enqueue(stream_);
        } catch (...) {                                     // never leave the stream in capture mode
            hipGraph_t partial = nullptr;
            (void)hipStreamEndCapture(stream_, &partial);
            if (partial) (void)hipGraphDestroy(partial);
            throw;
        }
        check_hip(hipStreamEndCapture(stream_, &graph_), "end capture");
        check_hip(hipGraphInstantiate(&exec_, graph_, nullptr, nullptr, 0), "graph instantiate");
    }
    // ... replay per frame pair (asynchronous on stream())
    void launch() {
        if (!exec_) capture();
        check_hip(hipGraphLaunch(exec_, stream_), "graph launch");
    }
    void synchronize() const { check_hip(hipStreamSynchronize(stream_), "synchronize"); }

  private:
    void drop_graph() {
        if (exec_) { (void)hipGraphExecDestroy(exec_); exec_ = nullptr; }
        if (graph_) { (void)hipGraphDestroy(graph_); graph_ = nullptr; }
    }
    void release() {
        drop_graph();
        for (auto &lv : levels_)
            for (void **p : {&lv.f1, &lv.f2, &lv.flow, &lv.warped, &lv.cost})
                if (*p) { (void)hipFree(*p); *p = nullptr; }
        if (stream_) { (void)hipStreamDestroy(stream_); stream_ = nullptr; }
    }
    int dtype_;
    CorrelationLayer corr_;
    FlowWarpLayer warp_;
    std::vector<PyramidLevel> levels_;
    hipStream_t stream_ = nullptr;
    hipGraph_t graph_ = nullptr;
    hipGraphExec_t exec_ = nullptr;
};

}  // namespace cerberus_rt
