#include "Layer.hpp"

#include <stdexcept>

namespace currennt_hip {
namespace layers {

namespace {
float jsonBias(const json::Value &layerChild) { return layerChild.hasMember("bias") ? (float)layerChild["bias"].getDouble() : 0.f; }
}

// ---- Layer ----------------------------------------------------------------------------------
Layer::Layer(cn_ctx *ctx, const json::Value &layerChild, cn_layer_kind kind, Layer *precedingLayer,
             int parallelSequences, int maxSeqLength, float bias)
    : m_ctx(ctx), m_handle(0)
    , m_name(layerChild.hasMember("name") ? layerChild["name"].getString() : "")
    , m_size(layerChild.hasMember("size") ? layerChild["size"].getInt() : 0)
    , m_parallelSequences(parallelSequences), m_maxSeqLength(maxSeqLength)
    , m_curMaxSeqLength(0), m_curMinSeqLength(0), m_curNumSeqs(0)
{
    if (!layerChild.hasMember("name")) throw std::runtime_error("Missing value 'name' in layer description");       // Layer.cpp:52-53
    if (m_name.empty()) throw std::runtime_error("Empty layer name in layer description");                           // :54-55
    if (!layerChild.hasMember("size")) throw std::runtime_error("Missing value 'size' in layer '" + m_name + "'");  // :56-57
    hipCheck(cn_layer_create(ctx, kind, precedingLayer ? precedingLayer->handle() : 0, m_size, bias,
                             precedingLayer ? 0 : parallelSequences, precedingLayer ? 0 : maxSeqLength, &m_handle), ctx);
}

Layer::~Layer() {}     // device memory belongs to the context (cn_ctx_destroy)

Hip::real_vector Layer::read(cn_buffer which, int dir, size_t count) const
{
    Hip::real_vector v(count);
    if (count) hipCheck(cn_layer_read(m_handle, which, dir, v.data(), count), m_ctx);
    return v;
}
Hip::real_vector Layer::outputs() const { return read(CN_BUF_OUTPUTS, 0, (size_t)m_curMaxSeqLength * m_parallelSequences * m_size); }
Hip::real_vector Layer::outputErrors() const { return read(CN_BUF_OUTPUT_ERRORS, 0, (size_t)m_curMaxSeqLength * m_parallelSequences * m_size); }

void Layer::loadSequences(const data_sets::DataSetFraction &fraction)
{
    m_curMaxSeqLength = fraction.maxSeqLength();
    m_curMinSeqLength = fraction.minSeqLength();
    m_curNumSeqs = fraction.numSequences();
    m_patTypes = fraction.patTypes();
}
void Layer::computeForwardPass() { hipCheck(cn_layer_forward(m_handle), m_ctx); }
void Layer::computeBackwardPass() { hipCheck(cn_layer_backward(m_handle), m_ctx); }

void Layer::exportLayer(json::Value *layersArray) const
{
    if (!layersArray->isArray()) throw std::runtime_error("The JSON value is not an array");
    json::Value o(json::Value::Object);
    o.addMember("name", name());
    o.addMember("type", type());
    o.addMember("size", size());
    layersArray->pushBack(o);
}

// ---- InputLayer -----------------------------------------------------------------------------
InputLayer::InputLayer(cn_ctx *ctx, const json::Value &layerChild, int parallelSequences, int maxSeqLength)
    : Layer(ctx, layerChild, CN_LAYER_INPUT, 0, parallelSequences, maxSeqLength, 0.f) {}
const std::string &InputLayer::type() const { static const std::string s("input"); return s; }
void InputLayer::loadSequences(const data_sets::DataSetFraction &fraction)
{
    if (fraction.inputPatternSize() != size())                                                   // InputLayer.cpp:52-55
        throw std::runtime_error("Input layer size of " + std::to_string(size()) + " != data input pattern size of " +
                                 std::to_string(fraction.inputPatternSize()));
    Layer::loadSequences(fraction);
}

// ---- TrainableLayer -------------------------------------------------------------------------
TrainableLayer::TrainableLayer(cn_ctx *ctx, const json::Value &layerChild, const json::Value *weightsSection, cn_layer_kind kind,
                               int inputWeightsPerBlock, int internalWeightsPerBlock, Layer &precedingLayer)
    : Layer(ctx, layerChild, kind, &precedingLayer, precedingLayer.parallelSequences(), precedingLayer.maxSeqLength(), jsonBias(layerChild))
    , m_precedingLayer(precedingLayer)
    , m_inputWeightsPerBlock(inputWeightsPerBlock), m_internalWeightsPerBlock(internalWeightsPerBlock)
    , m_bias(jsonBias(layerChild))
    , m_learningRate(layerChild.hasMember("learningRate") ? (real_t)layerChild["learningRate"].getDouble() : -1)
{
    if (!layerChild.hasMember("bias")) throw std::runtime_error("Missing value 'bias' in layer '" + name() + "'");   // TrainableLayer.cu:61-62
    // the layer's own learning rate (TrainableLayer.cu:58) is also what an armed update (cn_ctx_arm_update) applies to it
    if (m_learningRate >= 0) cn_layer_set_learning_rate(m_handle, m_learningRate);
    if (weightsSection && weightsSection->hasMember(name())) {                                                    // :68-101
        const json::Value &w = (*weightsSection)[name()];
        if (!w.isObject()) throw std::runtime_error("Weights section for layer '" + name() + "' is not an object");
        static const char *keys[3] = {"input", "bias", "internal"};
        for (int k = 0; k < 3; ++k)
            if (!w.hasMember(keys[k]) || !w[keys[k]].isArray())
                throw std::runtime_error("Missing array 'weights/" + name() + "/" + keys[k] + "'");
        const size_t P = (size_t)m_precedingLayer.size();
        if (w["input"].size() != (size_t)size() * inputWeightsPerBlock * P) throw std::runtime_error("Invalid number of input weights for layer '" + name() + "'");
        if (w["bias"].size() != (size_t)size() * inputWeightsPerBlock) throw std::runtime_error("Invalid number of bias weights for layer '" + name() + "'");
        if (w["internal"].size() != (size_t)size() * internalWeightsPerBlock) throw std::runtime_error("Invalid number of internal weights for layer '" + name() + "'");
        Hip::real_vector flat;
        flat.reserve(w["input"].size() + w["bias"].size() + w["internal"].size());
        for (int k = 0; k < 3; ++k)
            for (size_t i = 0; i < w[keys[k]].size(); ++i) flat.push_back((real_t)w[keys[k]][i].getDouble());
        setWeights(flat);
    }
    // otherwise NeuralNetwork draws the initial weights (TrainableLayer.cu:103-126)
}
int TrainableLayer::weightCount() const { return cn_layer_weight_count(m_handle); }
Hip::real_vector TrainableLayer::weights() const { return read(CN_BUF_WEIGHTS, 0, (size_t)weightCount()); }
Hip::real_vector TrainableLayer::weightUpdates() const { return read(CN_BUF_WEIGHT_UPDATES, 0, (size_t)weightCount()); }
void TrainableLayer::setWeights(const Hip::real_vector &w) { hipCheck(cn_layer_set_weights(m_handle, w.data(), (int)w.size()), m_ctx); }

void TrainableLayer::exportWeights(json::Value *weightsObject) const
{
    if (!weightsObject->isObject()) throw std::runtime_error("The JSON value is not an object");
    const Hip::real_vector w = weights();
    if (w.empty()) return;
    const size_t nIn = (size_t)size() * m_inputWeightsPerBlock * m_precedingLayer.size();
    const size_t nBias = (size_t)size() * m_inputWeightsPerBlock;
    json::Value in(json::Value::Array), bi(json::Value::Array), it(json::Value::Array);
    in.reserve(nIn); bi.reserve(nBias); it.reserve(w.size() - nIn - nBias);
    for (size_t i = 0; i < nIn; ++i) in.pushBack((double)w[i]);
    for (size_t i = 0; i < nBias; ++i) bi.pushBack((double)w[nIn + i]);
    for (size_t i = nIn + nBias; i < w.size(); ++i) it.pushBack((double)w[i]);
    json::Value sec(json::Value::Object);
    sec.addMember("input", in); sec.addMember("bias", bi); sec.addMember("internal", it);
    weightsObject->addMember(name(), sec);
}
void TrainableLayer::exportLayer(json::Value *layersArray) const
{
    Layer::exportLayer(layersArray);
    (*layersArray)[layersArray->size() - 1].addMember("bias", (double)m_bias);
}

// ---- FeedForward / Softmax / Lstm -----------------------------------------------------------
FeedForwardLayer::FeedForwardLayer(cn_ctx *ctx, const json::Value &layerChild, const json::Value *weightsSection, Layer &precedingLayer, cn_layer_kind kind)
    : TrainableLayer(ctx, layerChild, weightsSection, kind, 1, 0, precedingLayer)
    , m_type(kind == CN_LAYER_FF_TANH ? "feedforward_tanh" : (kind == CN_LAYER_FF_LOGISTIC ? "feedforward_logistic" : "feedforward_identity")) {}
const std::string &FeedForwardLayer::type() const { return m_type; }

SoftmaxLayer::SoftmaxLayer(cn_ctx *ctx, const json::Value &layerChild, const json::Value *weightsSection, Layer &precedingLayer)
    : TrainableLayer(ctx, layerChild, weightsSection, CN_LAYER_SOFTMAX, 1, 0, precedingLayer) {}
const std::string &SoftmaxLayer::type() const { static const std::string s("softmax"); return s; }

LstmLayer::LstmLayer(cn_ctx *ctx, const json::Value &layerChild, const json::Value *weightsSection, Layer &precedingLayer, bool bidirectional)
    : TrainableLayer(ctx, layerChild, weightsSection, bidirectional ? CN_LAYER_BLSTM : CN_LAYER_LSTM, 4,
                     (bidirectional ? 2 : 4) * (layerChild.hasMember("size") ? layerChild["size"].getInt() : 0) + 3, precedingLayer)   // LstmLayer.cu:525
    , m_isBidirectional(bidirectional) {}
const std::string &LstmLayer::type() const
{
    static const std::string su("lstm"), sb("blstm");
    return m_isBidirectional ? sb : su;
}
Hip::real_vector LstmLayer::internal(cn_buffer which, int dir) const
{
    const int H = size() / (m_isBidirectional ? 2 : 1);
    return read(which, dir, (size_t)curMaxSeqLength() * parallelSequences() * H);
}

// ---- post output layers ---------------------------------------------------------------------
PostOutputLayer::PostOutputLayer(cn_ctx *ctx, const json::Value &layerChild, cn_layer_kind kind, Layer &precedingLayer)
    : Layer(ctx, layerChild, kind, &precedingLayer, precedingLayer.parallelSequences(), precedingLayer.maxSeqLength(), 0.f)
    , m_precedingLayer(precedingLayer) {}
void PostOutputLayer::loadSequences(const data_sets::DataSetFraction &fraction)
{
    if (fraction.outputPatternSize() != size())                                                  // PostOutputLayer.cpp:70-73
        throw std::runtime_error("Output layer size of " + std::to_string(size()) + " != data target pattern size of " +
                                 std::to_string(fraction.outputPatternSize()));
    Layer::loadSequences(fraction);
}
real_t PostOutputLayer::calculateError()
{
    float e = 0; int c = 0;
    hipCheck(cn_loss_eval(m_handle, &e, &c), m_ctx);
    return e;
}

SsePostOutputLayer::SsePostOutputLayer(cn_ctx *ctx, const json::Value &layerChild, Layer &precedingLayer)
    : PostOutputLayer(ctx, layerChild, CN_LAYER_SSE, precedingLayer) {}
const std::string &SsePostOutputLayer::type() const { static const std::string s("sse"); return s; }

MulticlassClassificationLayer::MulticlassClassificationLayer(cn_ctx *ctx, const json::Value &layerChild, Layer &precedingLayer)
    : PostOutputLayer(ctx, layerChild, CN_LAYER_MULTICLASS_CLASSIFICATION, precedingLayer) {}
const std::string &MulticlassClassificationLayer::type() const { static const std::string s("multiclass_classification"); return s; }
int MulticlassClassificationLayer::countCorrectClassifications()
{
    float e = 0; int c = 0;
    hipCheck(cn_loss_eval(m_handle, &e, &c), m_ctx);
    return c;
}

#define CN_POST_LAYER(CLASS, KIND, TYPE)                                                                         \
    CLASS::CLASS(cn_ctx *ctx, const json::Value &layerChild, Layer &precedingLayer)                              \
        : PostOutputLayer(ctx, layerChild, KIND, precedingLayer) {}                                              \
    const std::string &CLASS::type() const { static const std::string s(TYPE); return s; }
CN_POST_LAYER(WeightedSsePostOutputLayer, CN_LAYER_WEIGHTEDSSE, "weightedsse")
CN_POST_LAYER(SseMaskPostOutputLayer, CN_LAYER_SSE_MASK, "wf")
CN_POST_LAYER(CePostOutputLayer, CN_LAYER_CE, "ce")
CN_POST_LAYER(RmsePostOutputLayer, CN_LAYER_RMSE, "rmse")
CN_POST_LAYER(BinaryClassificationLayer, CN_LAYER_BINARY_CLASSIFICATION, "binary_classification")
#undef CN_POST_LAYER
int BinaryClassificationLayer::countCorrectClassifications()
{
    float e = 0; int c = 0;
    hipCheck(cn_loss_eval(m_handle, &e, &c), m_ctx);
    return c;
}

}  // namespace layers
}  // namespace currennt_hip
