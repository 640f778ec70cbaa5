// cerberus_run -- command-line driver of cerberus_runtime.hpp: the flow pyramid's warp -> correlation
// (+ LeakyReLU) sequence from RAW buffers, no PyTorch, no Python: what main.cpp / cerberus.cpp:317-323 of the
// reference's TensorRT runtime are to its plugins, for the two ops this package owns.
//
//   cerberus_run --dir D --levels "C,H,W;C,H,W;..." [--batch N] [--dtype f32|f16|bf16] [--slope 0.1]
//                [--pad border|zeros] [--reps R] [--no-graph]
//
// Reads   D/f1_<l>.bin, D/f2_<l>.bin (N*C*H*W elements, NCHW) and, for l > 0, D/flow_<l>.bin (N*2*H*W),
// writes  D/cost_<l>.bin (N*81*H*W) and, for l > 0, D/warped_<l>.bin; prints one JSON line with the
// per-frame-pair latency of the hipGraph replay (HIP events on the launch stream).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "cerberus_runtime.hpp"

using namespace cerberus_rt;

static std::vector<char> read_file(const std::string &path, size_t bytes) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<char> buf(bytes);
    f.read(buf.data(), static_cast<std::streamsize>(bytes));
    if (static_cast<size_t>(f.gcount()) != bytes) throw std::runtime_error(path + ": expected " + std::to_string(bytes) + " bytes");
    return buf;
}
static void write_file(const std::string &path, const void *data, size_t bytes) {
    std::ofstream f(path, std::ios::binary);
    f.write(static_cast<const char *>(data), static_cast<std::streamsize>(bytes));
    if (!f) throw std::runtime_error("cannot write " + path);
}
static void upload(void *dst, const std::string &path, size_t bytes) {
    const auto buf = read_file(path, bytes);
    check_hip(hipMemcpy(dst, buf.data(), bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
}
static void download(const std::string &path, const void *src, size_t bytes) {
    std::vector<char> buf(bytes);
    check_hip(hipMemcpy(buf.data(), src, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    write_file(path, buf.data(), bytes);
}

int main(int argc, char **argv) {
    std::string dir = ".", levels_s, dtype_s = "f32", pad_s = "border";
    int batch = 1, reps = 20;
    float slope = 0.1f;
    bool use_graph = true;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) throw std::invalid_argument(a + " needs a value"); return argv[++i]; };
        try {
            if (a == "--dir") dir = next();
            else if (a == "--levels") levels_s = next();
            else if (a == "--batch") batch = std::stoi(next());
            else if (a == "--dtype") dtype_s = next();
            else if (a == "--slope") slope = std::stof(next());
            else if (a == "--pad") pad_s = next();
            else if (a == "--reps") reps = std::stoi(next());
            else if (a == "--no-graph") use_graph = false;
            else if (a == "--abi") { std::printf("%d\n", cerberus_abi_version()); return 0; }
            else { std::fprintf(stderr, "usage: cerberus_run --dir D --levels \"C,H,W;...\" [--batch N] [--dtype f32|f16|bf16] "
                                        "[--slope 0.1] [--pad border|zeros] [--reps R] [--no-graph] | --abi\n"); return a == "--help" ? 0 : 2; }
        } catch (const std::exception &e) { std::fprintf(stderr, "cerberus_run: %s\n", e.what()); return 2; }
    }
    try {
        const int dtype = dtype_s == "f32" ? CERB_F32 : dtype_s == "f16" ? CERB_F16 : dtype_s == "bf16" ? CERB_BF16 : -1;
        if (dtype < 0) throw std::invalid_argument("--dtype must be f32, f16 or bf16");
        const int pad = pad_s == "border" ? CERB_PAD_BORDER : pad_s == "zeros" ? CERB_PAD_ZEROS : -1;
        if (pad < 0) throw std::invalid_argument("--pad must be border or zeros");
        if (reps < 1 || batch < 1) { std::fprintf(stderr, "cerberus_run: --reps and --batch must be >= 1\n"); return 2; }
        std::vector<Dims4> levels;
        std::stringstream ss(levels_s);
        for (std::string item; std::getline(ss, item, ';');) {
            Dims4 d; d.n = batch;
            if (std::sscanf(item.c_str(), "%d,%d,%d", &d.c, &d.h, &d.w) != 3) throw std::invalid_argument("bad --levels item: " + item);
            levels.push_back(d);
        }
        if (levels.empty()) throw std::invalid_argument("--levels is empty");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device (there is no CPU path)");

        const CorrelationLayer corr(4, 1, 4, 1, 1, 1, slope);     // configs/HRNetV2_kt.json:77-84
        const FlowWarpLayer warp(pad, CERB_INTERP_BILINEAR);
        FlowPyramidGraph pyr(levels, dtype, corr, warp);
        const size_t e = dtype_bytes(dtype);
        for (size_t l = 0; l < pyr.levels().size(); ++l) {
            auto &lv = pyr.levels()[l];
            const std::string s = std::to_string(l);
            upload(lv.f1, dir + "/f1_" + s + ".bin", size_t(lv.feat.count()) * e);
            upload(lv.f2, dir + "/f2_" + s + ".bin", size_t(lv.feat.count()) * e);
            if (lv.has_flow) upload(lv.flow, dir + "/flow_" + s + ".bin", size_t(lv.feat.n) * 2 * lv.feat.h * lv.feat.w * e);
        }
        if (use_graph) pyr.capture();
        auto run = [&]() { if (use_graph) pyr.launch(); else pyr.enqueue(pyr.stream()); };
        for (int i = 0; i < 3; ++i) run();
        pyr.synchronize();
        hipEvent_t a, b;
        check_hip(hipEventCreate(&a), "event"); check_hip(hipEventCreate(&b), "event");
        check_hip(hipEventRecord(a, pyr.stream()), "record");
        for (int i = 0; i < reps; ++i) run();
        check_hip(hipEventRecord(b, pyr.stream()), "record");
        check_hip(hipEventSynchronize(b), "sync");
        float ms = 0.f;
        check_hip(hipEventElapsedTime(&ms, a, b), "elapsed");
        for (size_t l = 0; l < pyr.levels().size(); ++l) {
            auto &lv = pyr.levels()[l];
            const std::string s = std::to_string(l);
            download(dir + "/cost_" + s + ".bin", lv.cost, size_t(lv.cost_dims.count()) * e);
            if (lv.has_flow) download(dir + "/warped_" + s + ".bin", lv.warped, size_t(lv.feat.count()) * e);
        }
        std::printf("{\"runtime\": \"cerberus_run\", \"abi\": %d, \"levels\": %zu, \"batch\": %d, \"dtype\": \"%s\", "
                    "\"launch\": \"%s\", \"reps\": %d, \"us_per_frame_pair_direction\": %.2f}\n",
                    cerberus_abi_version(), levels.size(), batch, dtype_s.c_str(), use_graph ? "hipGraph" : "eager", reps,
                    1e3 * ms / reps);
    } catch (const std::exception &ex) {
        std::fprintf(stderr, "cerberus_run: %s\n", ex.what());
        return 1;
    }
    return 0;
}
