// pnn_tf_compat.h -- header-only look-alike of the TensorFlow C++ API subset that the reference's HM side uses
// (SURVEY.md Appendix C), implemented on the C ABI of pnn_hip.h.  With `-I<repo>/include/tf_compat` in front
// of the include path (it also shadows python2.7/Python.h, the other dependency of TComPrediction.h:45-47), the
// reference's hevc/hm_common/c++/source_common/*.cpp and both modified HM-16.15 trees compile UNCHANGED and link with
// libpnn_hip.so alone -- tools/hm/Makefile does exactly that, tests/test_hm.py runs the resulting encoders / decoders:
//
//   tensorflow::Tensor(DT_FLOAT, {1, 80}).flat<float>().data() / .dims() / .shape().dim_size(i)
//   tensorflow::GraphDef + ReadBinaryProto(Env::Default(), path, &graph_def)   -> remembers the model path
//   tensorflow::NewSession(SessionOptions()) ; session->Create(graph_def)      -> loads the model (.pnnw)
//   session->Run({{"node_flattened_context", T}}, {"fully_connected/node_output"}, {}, &out)
//   session->Run({{"node_portion_above", A}, {"node_portion_left", L}}, {".../node_output"}, {}, &out)
//   tensorflow::Status (.ok(), Status::OK(), operator<<), errors::NotFound(...), LOG(ERROR), tensorflow::string
//
// Reference call sites: integration_prediction_neural_network.cpp:3-69, TComPrediction.cpp:564-622,
// TComPattern.cpp:344-360.  Paths listed in the model table must point at `.pnnw` files.
//
// HM constructs its sessions with default SessionOptions, so a deployment is configured through the environment:
//   PNN_DEVICE=<n>            HIP device of this process's sessions (default 0)
//   PNN_SERVICE_SOCKET=<path> do not own a GPU context: send every Run to the batching service at that socket
//                             (include/pnn_service.h) -- many encoder processes then share one GPU in batched launches
//   PNN_CACHE_MB=<n>          per-session prediction cache (repeated identical Runs of HM's RD search), default 64
//   PNN_STATS=1               print Run / cache-hit counts per session on stderr when the session is destroyed
//   PNN_ASYNC_LOAD=0          read / pack / upload the weights inside Session::Create (default: in a thread that the first Run joins)
// Sessions always run with canonical_order = 1: a block's prediction does not depend on the batch it travels in, so an
// encoder behind the batching service and a stand-alone decoder reconstruct the same picture.
#ifndef PNN_TF_COMPAT_H
#define PNN_TF_COMPAT_H

#include "pnn_hip.h"
#include "pnn_service.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <utility>
#include <vector>
#include <thread>

namespace tensorflow {

typedef std::string string;
typedef long long int64;

enum DataType { DT_INVALID = 0, DT_FLOAT = 1 };

class Status {
public:
    Status() : ok_(true) {}
    Status(bool ok, const std::string& msg) : ok_(ok), msg_(msg) {}
    static Status OK() { return Status(); }
    bool ok() const { return ok_; }
    const std::string& error_message() const { return msg_; }
    std::string ToString() const { return ok_ ? "OK" : msg_; }
private:
    bool ok_;
    std::string msg_;
};
inline std::ostream& operator<<(std::ostream& os, const Status& s) { return os << s.ToString(); }

namespace errors {
inline void pnn_append(std::ostringstream&) {}
template <typename T, typename... Rest>
inline void pnn_append(std::ostringstream& os, const T& v, const Rest&... rest) { os << v; pnn_append(os, rest...); }
template <typename... Args>
inline Status NotFound(const Args&... args) { std::ostringstream os; pnn_append(os, args...); return Status(false, "Not found: " + os.str()); }
template <typename... Args>
inline Status InvalidArgument(const Args&... args) { std::ostringstream os; pnn_append(os, args...); return Status(false, "Invalid argument: " + os.str()); }
template <typename... Args>
inline Status Internal(const Args&... args) { std::ostringstream os; pnn_append(os, args...); return Status(false, "Internal: " + os.str()); }
template <typename... Args>
inline Status FailedPrecondition(const Args&... args) { std::ostringstream os; pnn_append(os, args...); return Status(false, "Failed precondition: " + os.str()); }
}  // namespace errors

class TensorShape {
public:
    TensorShape() {}
    TensorShape(std::initializer_list<int64> d) : dims_(d) {}
    explicit TensorShape(const std::vector<int64>& d) : dims_(d) {}
    int dims() const { return (int)dims_.size(); }
    int64 dim_size(int i) const { return dims_.at(i); }
    int64 num_elements() const { int64 n = 1; for (int64 d : dims_) n *= d; return n; }
private:
    std::vector<int64> dims_;
};

class Tensor {
public:
    Tensor() : dtype_(DT_INVALID) {}
    Tensor(DataType dt, const TensorShape& shape) : dtype_(dt), shape_(shape), buf_(new std::vector<float>((size_t)shape.num_elements(), 0.f)) {}
    template <typename T>
    struct Flat {
        T* ptr; int64 n;
        T* data() const { return ptr; }
        int64 size() const { return n; }
        T& operator()(int64 i) const { return ptr[i]; }
    };
    template <typename T> Flat<T> flat() { return Flat<T>{buf_ ? buf_->data() : nullptr, shape_.num_elements()}; }
    template <typename T> Flat<const T> flat() const { return Flat<const T>{buf_ ? buf_->data() : nullptr, shape_.num_elements()}; }
    int dims() const { return shape_.dims(); }
    const TensorShape& shape() const { return shape_; }
    int64 dim_size(int i) const { return shape_.dim_size(i); }
    int64 NumElements() const { return shape_.num_elements(); }
    DataType dtype() const { return dtype_; }
private:
    DataType dtype_;
    TensorShape shape_;
    std::shared_ptr<std::vector<float> > buf_;     // copies share the buffer, as TF tensors do
};

class Env {
public:
    static Env* Default() { static Env e; return &e; }
};

// The "graph" is the path of the model file; Session::Create loads it.
class GraphDef {
public:
    std::string pnn_model_path;
};

inline Status ReadBinaryProto(Env*, const std::string& path, GraphDef* graph_def)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return errors::NotFound(path, "; No such file or directory");
    std::fclose(f);
    graph_def->pnn_model_path = path;
    return Status::OK();
}

struct SessionOptions {
    int pnn_device = -1;         // HIP device index; -1: $PNN_DEVICE, else 0
    float pnn_mean = 0.f;        // only used by the fused Pel entry points; the float Run() path never adds the mean
    long pnn_cache_mb = -1;      // -1: $PNN_CACHE_MB, else 64
};

class Session {
public:
    virtual ~Session() {}
    virtual Status Create(const GraphDef& graph) = 0;
    virtual Status Run(const std::vector<std::pair<string, Tensor> >& inputs, const std::vector<string>& output_tensor_names,
                       const std::vector<string>& target_node_names, std::vector<Tensor>* outputs) = 0;
    virtual Status Close() { return Status::OK(); }
};

class PnnSession : public Session {
public:
    explicit PnnSession(const SessionOptions& o) : opts_(o), ctx_(nullptr), client_(nullptr), width_(0), is_fc_(0), runs_(0), load_rc_(PNN_OK) {}
    ~PnnSession() override
    {
        finish_load();
        if (std::getenv("PNN_STATS") && width_) {
            long hits = 0, misses = 0;
            if (ctx_) pnn_cache_stats(ctx_, &hits, &misses);
            if (client_) pnn_client_cache_stats(client_, &hits, &misses);
            std::fprintf(stderr, "[pnn] session width %d (%s%s): %ld Run calls, %ld answered from the cache\n", width_,
                         is_fc_ ? "fully-connected" : "convolutional", client_ ? ", via service" : "", runs_, hits);
        }
        if (ctx_) pnn_destroy(ctx_);
        if (client_) pnn_client_close(client_);
    }
    pnn_ctx* pnn_context() { finish_load(); return ctx_; }

    Status Create(const GraphDef& graph) override
    {
        finish_load();
        if (ctx_) { pnn_destroy(ctx_); ctx_ = nullptr; }
        if (client_) { pnn_client_close(client_); client_ = nullptr; }
        width_ = 0;
        // the model's header (width, kind): what a missing or foreign file costs is paid here, like a frozen graph that does not parse
        struct { char magic[4]; uint32_t version, width, is_fc; } h;
        FILE* f = std::fopen(graph.pnn_model_path.c_str(), "rb");
        if (!f) return errors::NotFound(graph.pnn_model_path, "; No such file or directory");
        const bool ok = std::fread(&h, sizeof h, 1, f) == 1 && !std::memcmp(h.magic, "PNNW", 4);
        std::fclose(f);
        if (!ok) return errors::InvalidArgument(graph.pnn_model_path, " is not a PNNW file");
        // The arithmetic contract: an encoder and the decoder of its bitstream must predict with the same last float
        // bits.  $PNN_EXPECT_TAG = the tag the other side reported (pnn_arithmetic_tag; tools/hm/campaign.py records and passes it):
        // this session fails Create() when whatever will answer its Run() calls -- the batching service or the local context --
        // computes on another arithmetic or summation order, instead of decoding a drifting picture.  $PNN_PRINT_TAG=1 prints it.
        const char* expect = std::getenv("PNN_EXPECT_TAG");
        if (const char* sock = std::getenv("PNN_SERVICE_SOCKET")) {   // served remotely: nothing else is needed here
            if (pnn_client_connect(&client_, sock) != PNN_OK) return errors::Internal("no PNN batching service at ", sock);
            width_ = (int)h.width; is_fc_ = (int)h.is_fc;
            if (expect || std::getenv("PNN_PRINT_TAG")) {
                char tag[256];
                if (pnn_client_arithmetic_tag(client_, width_, tag, sizeof tag) != PNN_OK) return errors::Internal("the PNN batching service at ", sock, " does not report its arithmetic tag");
                if (std::getenv("PNN_PRINT_TAG")) std::fprintf(stderr, "[pnn] arithmetic tag, width %d (service): %s\n", width_, tag);
                if (expect && std::strcmp(expect, tag))
                    return errors::FailedPrecondition("arithmetic mismatch for width ", width_, ": the batching service computes on \"", tag, "\", expected \"", expect, "\"");
            }
            return Status::OK();
        }
        int dev = opts_.pnn_device;
        if (dev < 0) { const char* e = std::getenv("PNN_DEVICE"); dev = e ? std::atoi(e) : 0; }
        int rc = pnn_create_empty(&ctx_, opts_.pnn_mean, dev);       // no HIP device: fails HERE, loudly
        if (rc != PNN_OK) return errors::Internal(pnn_last_error(nullptr));
        long cache_mb = opts_.pnn_cache_mb;
        if (cache_mb < 0) { const char* e = std::getenv("PNN_CACHE_MB"); cache_mb = e ? std::atol(e) : 64; }
        width_ = (int)h.width; is_fc_ = (int)h.is_fc;
        if (expect || std::getenv("PNN_PRINT_TAG")) {                  // (the tag depends on the context's options, not on the model: no need to wait for the load)
            char tag[256];
            if (pnn_arithmetic_tag(ctx_, tag, sizeof tag) != PNN_OK) return errors::Internal("pnn_arithmetic_tag failed");
            if (std::getenv("PNN_PRINT_TAG")) std::fprintf(stderr, "[pnn] arithmetic tag, width %d (local context): %s\n", width_, tag);
            if (expect && std::strcmp(expect, tag))
                return errors::FailedPrecondition("arithmetic mismatch for width ", width_, ": this process computes on \"", tag, "\", expected \"", expect, "\"");
        }
        // Reading, packing and uploading the weights (0.02 ... 0.2 s per model, 0.5 s for HM's five) runs in a thread of its own:
        // HM creates its five sessions one after the other and needs none of them before the first intra block, so the five
        // loads overlap each other and the rest of the codec's start-up.  The first Run() (or the destructor) joins.
        // PNN_ASYNC_LOAD=0: load here, synchronously.
        const std::string path = graph.pnn_model_path;
        load_rc_ = PNN_OK;
        auto load = [this, path, cache_mb]() {
            load_rc_ = pnn_load_model_file(ctx_, path.c_str());
            if (load_rc_ != PNN_OK) { load_err_ = pnn_last_error(ctx_); return; }
            pnn_set_option(ctx_, "canonical_order", 1);
            pnn_set_option(ctx_, "cache_mb", cache_mb);
            int fc = 0;
            if (pnn_model_info(ctx_, width_, &fc, nullptr, nullptr) != PNN_OK || fc != is_fc_) {
                load_rc_ = PNN_E_MODEL; load_err_ = "model header and contents disagree in " + path;
            }
        };
        const char* as = std::getenv("PNN_ASYNC_LOAD");
        if (as && std::atoi(as) == 0) {
            load();
            if (load_rc_ != PNN_OK) return errors::InvalidArgument(load_err_);
        } else {
            loader_ = std::thread(load);
        }
        return Status::OK();
    }

    Status Run(const std::vector<std::pair<string, Tensor> >& inputs, const std::vector<string>& output_tensor_names,
               const std::vector<string>&, std::vector<Tensor>* outputs) override
    {
        if (!ctx_ && !client_) return errors::Internal("Session::Run before Session::Create");
        finish_load();
        if (load_rc_ != PNN_OK) return errors::InvalidArgument(load_err_);
        if (!outputs || output_tensor_names.size() != 1) return errors::InvalidArgument("exactly one fetch is supported");
        const Tensor* ctx = nullptr; const Tensor* above = nullptr; const Tensor* left = nullptr;
        for (const auto& kv : inputs) {
            if (kv.first == "node_flattened_context" || kv.first == "node_flattened_context:0") ctx = &kv.second;
            else if (kv.first == "node_portion_above" || kv.first == "node_portion_above:0") above = &kv.second;
            else if (kv.first == "node_portion_left" || kv.first == "node_portion_left:0") left = &kv.second;
            else return errors::NotFound("feed ", kv.first, " is not a placeholder of the PNN graph");
        }
        const string& fetch = output_tensor_names[0];
        const int w = width_;
        const int64 w2 = (int64)w * w;
        int rc = PNN_OK;
        ++runs_;
        if (is_fc_) {
            if (!ctx || fetch.find("fully_connected/node_output") != 0) return errors::NotFound("FetchOutputs node ", fetch, ": not found");
            if (ctx->NumElements() % (5 * w2)) return errors::InvalidArgument("node_flattened_context must be [N, ", 5 * w2, "]");
            const int n = (int)(ctx->NumElements() / (5 * w2));
            Tensor out(DT_FLOAT, TensorShape({n, w, w, 1}));
            const float* x = ctx->flat<float>().data();
            float* y = out.flat<float>().data();
            if (client_) for (int i = 0; i < n && rc == PNN_OK; i++) rc = pnn_client_predict_f32(client_, w, x + i * 5 * w2, nullptr, y + i * w2);
            else rc = pnn_predict_fc(ctx_, w, x, n, y);
            if (rc != PNN_OK) return errors::Internal(client_ ? "the PNN batching service failed" : pnn_last_error(ctx_));
            outputs->assign(1, out);
        } else if (ctx && !above && !left && w <= 8) {
            // Extension: HM feeds widths 4 and 8 as ONE flattened tensor [above w x 3w | left 2w x w] (TComPattern.cpp:344-354)
            // because its production models for them are fully-connected; the reference also trains CONVOLUTIONAL nets for
            // these widths (pnn/results/width_target_{4,8}/convolutional/..., the only trained weights it ships).  A table
            // entry may point at such a model: the flattened tensor is simply its two portions back to back.
            if (ctx->NumElements() % (5 * w2)) return errors::InvalidArgument("node_flattened_context must be [N, ", 5 * w2, "]");
            const int n = (int)(ctx->NumElements() / (5 * w2));
            Tensor out(DT_FLOAT, TensorShape({n, w, w, 1}));
            const float* x = ctx->flat<float>().data();
            float* y = out.flat<float>().data();
            for (int i = 0; i < n && rc == PNN_OK; i++)
                rc = client_ ? pnn_client_predict_f32(client_, w, x + i * 5 * w2, x + i * 5 * w2 + 3 * w2, y + i * w2)
                             : pnn_predict_conv(ctx_, w, x + i * 5 * w2, x + i * 5 * w2 + 3 * w2, 1, y + i * w2);
            if (rc != PNN_OK) return errors::Internal(client_ ? "the PNN batching service failed" : pnn_last_error(ctx_));
            outputs->assign(1, out);
        } else {
            if (!above || !left || fetch.find("convolutional/merger/transpose_convolution_") != 0)
                return errors::NotFound("FetchOutputs node ", fetch, ": not found");
            if (above->NumElements() % (3 * w2) || left->NumElements() / (2 * w2) != above->NumElements() / (3 * w2))
                return errors::InvalidArgument("node_portion_above / node_portion_left have the wrong shape for width ", w);
            const int n = (int)(above->NumElements() / (3 * w2));
            Tensor out(DT_FLOAT, TensorShape({n, w, w, 1}));
            const float* a = above->flat<float>().data();
            const float* l = left->flat<float>().data();
            float* y = out.flat<float>().data();
            if (client_) for (int i = 0; i < n && rc == PNN_OK; i++) rc = pnn_client_predict_f32(client_, w, a + i * 3 * w2, l + i * 2 * w2, y + i * w2);
            else rc = pnn_predict_conv(ctx_, w, a, l, n, y);
            if (rc != PNN_OK) return errors::Internal(client_ ? "the PNN batching service failed" : pnn_last_error(ctx_));
            outputs->assign(1, out);
        }
        return Status::OK();
    }

private:
    void finish_load() { if (loader_.joinable()) loader_.join(); }
    SessionOptions opts_;
    pnn_ctx* ctx_;
    pnn_client* client_;
    int width_, is_fc_;
    long runs_;
    std::thread loader_;           // the model load started by Create()
    int load_rc_;
    std::string load_err_;
};

inline Session* NewSession(const SessionOptions& options) { return new PnnSession(options); }

// LOG(ERROR) << ...  (tensorflow/core/platform/logging.h)
class PnnLogLine {
public:
    explicit PnnLogLine(const char* sev) { os_ << sev << " "; }
    ~PnnLogLine() { std::cerr << os_.str() << std::endl; }
    template <typename T> PnnLogLine& operator<<(const T& v) { os_ << v; return *this; }
private:
    std::ostringstream os_;
};

}  // namespace tensorflow

#ifndef LOG
#define PNN_LOG_INFO ::tensorflow::PnnLogLine("I")
#define PNN_LOG_WARNING ::tensorflow::PnnLogLine("W")
#define PNN_LOG_ERROR ::tensorflow::PnnLogLine("E")
#define PNN_LOG_FATAL ::tensorflow::PnnLogLine("F")
#define LOG(severity) PNN_LOG_##severity
#endif

#endif  // PNN_TF_COMPAT_H
