// Test program for include/pnn_tf_compat.h: drives libpnn_hip.so exactly through the TensorFlow C++ calls the
// reference's HM side makes (TComPrediction.cpp:564-622 is the model for the sequence, this
// is not a copy of it).  usage: hm_callsite_sample <fc.pnnw> <width> <conv.pnnw> <width>
// Prints the predictions so that the Python test can compare them with the oracle.
#include "tensorflow/core/public/session.h"
#include "tensorflow/core/framework/tensor.h"
#include "tensorflow/core/platform/logging.h"

#include <cstdlib>
#include <memory>

static int run_one(const char* path, int w, bool fc)
{
    tensorflow::GraphDef graph_def;
    tensorflow::Status st = ReadBinaryProto(tensorflow::Env::Default(), path, &graph_def);
    if (!st.ok()) { LOG(ERROR) << st; return 1; }
    std::unique_ptr<tensorflow::Session> session(tensorflow::NewSession(tensorflow::SessionOptions()));
    st = session->Create(graph_def);
    if (!st.ok()) { LOG(ERROR) << st; return 1; }
    std::vector<tensorflow::Tensor> out;
    if (fc) {
        tensorflow::Tensor t(tensorflow::DT_FLOAT, {1, 5 * w * w});
        float* p = t.flat<float>().data();
        for (int i = 0; i < 5 * w * w; i++) p[i] = (float)((i * 37) % 256) - 117.8952234192841f;
        st = session->Run({{"node_flattened_context", t}}, {"fully_connected/node_output"}, {}, &out);
    } else {
        tensorflow::Tensor a(tensorflow::DT_FLOAT, {1, w, 3 * w, 1}), l(tensorflow::DT_FLOAT, {1, 2 * w, w, 1});
        float* pa = a.flat<float>().data();
        float* pl = l.flat<float>().data();
        for (int i = 0; i < 3 * w * w; i++) pa[i] = (float)((i * 37) % 256) - 117.8952234192841f;
        for (int i = 0; i < 2 * w * w; i++) pl[i] = (float)((i * 53 + 11) % 256) - 117.8952234192841f;
        const tensorflow::string name = w == 16 ? "convolutional/merger/transpose_convolution_3/node_output"
                                                : (w >= 32 ? "convolutional/merger/transpose_convolution_4/node_output"
                                                           : "convolutional/merger/transpose_convolution_1/node_output");
        st = session->Run({{"node_portion_above", a}, {"node_portion_left", l}}, {name}, {}, &out);
    }
    if (!st.ok()) { LOG(ERROR) << st; return 1; }
    const tensorflow::Tensor& pred = out.at(0);
    if (pred.dims() != 4 || pred.shape().dim_size(1) != w || pred.shape().dim_size(2) != w) return 2;
    const float* pp = pred.flat<float>().data();
    for (int i = 0; i < w * w; i++) printf("%.6f\n", pp[i]);
    // a wrong fetch name must fail like TF does, not silently succeed
    st = session->Run({}, {"no_such_node"}, {}, &out);
    return st.ok() ? 3 : 0;
}

int main(int argc, char** argv)
{
    if (argc != 5) return 64;
    if (int rc = run_one(argv[1], atoi(argv[2]), true)) return rc;
    printf("----\n");
    return run_one(argv[3], atoi(argv[4]), false);
}
