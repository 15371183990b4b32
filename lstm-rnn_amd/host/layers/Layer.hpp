// Layer objects of the host side: same class roles, method names and error texts as
// currennt_lib/src/layers/{Layer,TrainableLayer,InputLayer,LstmLayer,FeedForwardLayer,SoftmaxLayer,
// PostOutputLayer,SsePostOutputLayer,MulticlassClassificationLayer}.hpp for the `Hip` device policy.
// Every compute method is one call into the C ABI (include/currennt_hip.h); vectors returned by
// outputs()/weights()/... are host copies in the reference layouts.
#pragma once

#include <string>
#include <vector>

#include "../Json.hpp"
#include "../Types.hpp"
#include "../data_sets/DataSet.hpp"

namespace currennt_hip {
namespace layers {

class Layer {                                             // layers/Layer.hpp:40-179
public:
    Layer(cn_ctx *ctx, const json::Value &layerChild, cn_layer_kind kind, Layer *precedingLayer,
          int parallelSequences, int maxSeqLength, float bias);
    virtual ~Layer();

    const std::string &name() const { return m_name; }
    int size() const { return m_size; }
    int parallelSequences() const { return m_parallelSequences; }
    int maxSeqLength() const { return m_maxSeqLength; }
    int curMaxSeqLength() const { return m_curMaxSeqLength; }
    int curMinSeqLength() const { return m_curMinSeqLength; }
    int curNumSeqs() const { return m_curNumSeqs; }
    const Hip::pattype_vector &patTypes() const { return m_patTypes; }
    virtual const std::string &type() const = 0;
    virtual bool isTrainable() const { return false; }
    virtual bool isPostOutput() const { return false; }

    Hip::real_vector outputs() const;                     // Layer.hpp:132
    Hip::real_vector outputErrors() const;                // Layer.hpp:153

    virtual void loadSequences(const data_sets::DataSetFraction &fraction);   // Layer.cpp:134-141
    virtual void computeForwardPass();
    virtual void computeBackwardPass();
    virtual void exportLayer(json::Value *layersArray) const;                 // Layer.cpp:143-157

    cn_layer *handle() const { return m_handle; }
    cn_ctx *context() const { return m_ctx; }

protected:
    Hip::real_vector read(cn_buffer which, int dir, size_t count) const;

    cn_ctx *m_ctx;
    cn_layer *m_handle;
    std::string m_name;
    int m_size;
    int m_parallelSequences, m_maxSeqLength, m_curMaxSeqLength, m_curMinSeqLength, m_curNumSeqs;
    Hip::pattype_vector m_patTypes;
};

class InputLayer : public Layer {                         // layers/InputLayer.{hpp,cpp}
public:
    InputLayer(cn_ctx *ctx, const json::Value &layerChild, int parallelSequences, int maxSeqLength);
    const std::string &type() const;
    void loadSequences(const data_sets::DataSetFraction &fraction);
};

class TrainableLayer : public Layer {                     // layers/TrainableLayer.{hpp,cu}
public:
    TrainableLayer(cn_ctx *ctx, const json::Value &layerChild, const json::Value *weightsSection, cn_layer_kind kind,
                   int inputWeightsPerBlock, int internalWeightsPerBlock, Layer &precedingLayer);
    bool isTrainable() const { return true; }
    Layer &precedingLayer() { return m_precedingLayer; }
    const Layer &precedingLayer() const { return m_precedingLayer; }
    real_t bias() const { return m_bias; }
    real_t learningRate() const { return m_learningRate; }
    Hip::real_vector weights() const;                     // TrainableLayer.hpp:119
    Hip::real_vector weightUpdates() const;               // TrainableLayer.hpp:133
    void setWeights(const Hip::real_vector &w);
    int weightCount() const;
    void exportWeights(json::Value *weightsObject) const; // TrainableLayer.cu:211-248
    void exportLayer(json::Value *layersArray) const;     // TrainableLayer.cu:250-255

private:
    Layer &m_precedingLayer;
    int m_inputWeightsPerBlock, m_internalWeightsPerBlock;
    real_t m_bias, m_learningRate;
};

class FeedForwardLayer : public TrainableLayer {          // layers/FeedForwardLayer.{hpp,cu}
public:
    FeedForwardLayer(cn_ctx *ctx, const json::Value &layerChild, const json::Value *weightsSection, Layer &precedingLayer,
                     cn_layer_kind kind);
    const std::string &type() const;
private:
    std::string m_type;
};

class SoftmaxLayer : public TrainableLayer {              // layers/SoftmaxLayer.{hpp,cu}
public:
    SoftmaxLayer(cn_ctx *ctx, const json::Value &layerChild, const json::Value *weightsSection, Layer &precedingLayer);
    const std::string &type() const;
};

class LstmLayer : public TrainableLayer {                 // layers/LstmLayer.{hpp,cu}
public:
    LstmLayer(cn_ctx *ctx, const json::Value &layerChild, const json::Value *weightsSection, Layer &precedingLayer, bool bidirectional);
    const std::string &type() const;
    bool isBidirectional() const { return m_isBidirectional; }
    // the ten per-direction vectors of LstmLayer.hpp:170-233; unlike the reference they are also
    // available for bidirectional layers (direction 0 = forward states, 1 = backward states)
    Hip::real_vector cellStates(int dir = 0) const { return internal(CN_BUF_LSTM_CELL_STATES, dir); }
    Hip::real_vector netInputActs(int dir = 0) const { return internal(CN_BUF_LSTM_NI_ACTS, dir); }
    Hip::real_vector inputGateActs(int dir = 0) const { return internal(CN_BUF_LSTM_IG_ACTS, dir); }
    Hip::real_vector forgetGateActs(int dir = 0) const { return internal(CN_BUF_LSTM_FG_ACTS, dir); }
    Hip::real_vector outputGateActs(int dir = 0) const { return internal(CN_BUF_LSTM_OG_ACTS, dir); }
    Hip::real_vector netInputDeltas(int dir = 0) const { return internal(CN_BUF_LSTM_NI_DELTAS, dir); }
    Hip::real_vector inputGateDeltas(int dir = 0) const { return internal(CN_BUF_LSTM_IG_DELTAS, dir); }
    Hip::real_vector forgetGateDeltas(int dir = 0) const { return internal(CN_BUF_LSTM_FG_DELTAS, dir); }
    Hip::real_vector outputGateDeltas(int dir = 0) const { return internal(CN_BUF_LSTM_OG_DELTAS, dir); }
private:
    Hip::real_vector internal(cn_buffer which, int dir) const;
    bool m_isBidirectional;
};

class PostOutputLayer : public Layer {                    // layers/PostOutputLayer.{hpp,cpp}
public:
    PostOutputLayer(cn_ctx *ctx, const json::Value &layerChild, cn_layer_kind kind, Layer &precedingLayer);
    bool isPostOutput() const { return true; }
    void loadSequences(const data_sets::DataSetFraction &fraction);
    virtual real_t calculateError();                      // PostOutputLayer.hpp:80
    virtual int countCorrectClassifications() { return -1; }
protected:
    Layer &m_precedingLayer;
};

class SsePostOutputLayer : public PostOutputLayer {       // layers/SsePostOutputLayer.{hpp,cu}
public:
    SsePostOutputLayer(cn_ctx *ctx, const json::Value &layerChild, Layer &precedingLayer);
    const std::string &type() const;
};

class MulticlassClassificationLayer : public PostOutputLayer {   // layers/MulticlassClassificationLayer.{hpp,cu}
public:
    MulticlassClassificationLayer(cn_ctx *ctx, const json::Value &layerChild, Layer &precedingLayer);
    const std::string &type() const;
    int countCorrectClassifications();
};

// The remaining post output layers of LayerFactory.cu:52-87; each is a thin handle on the C ABI kind.
class WeightedSsePostOutputLayer : public PostOutputLayer {   // layers/WeightedSsePostOutputLayer.{hpp,cu}; size = 2 x output layer
public:
    WeightedSsePostOutputLayer(cn_ctx *ctx, const json::Value &layerChild, Layer &precedingLayer);
    const std::string &type() const;
};
class SseMaskPostOutputLayer : public PostOutputLayer {       // layers/SseMaskPostOutputLayer.{hpp,cu} ("wf"); size = 2 x output layer
public:
    SseMaskPostOutputLayer(cn_ctx *ctx, const json::Value &layerChild, Layer &precedingLayer);
    const std::string &type() const;
};
class CePostOutputLayer : public PostOutputLayer {            // layers/CePostOutputLayer.{hpp,cu}
public:
    CePostOutputLayer(cn_ctx *ctx, const json::Value &layerChild, Layer &precedingLayer);
    const std::string &type() const;
};
class RmsePostOutputLayer : public PostOutputLayer {          // layers/RmsePostOutputLayer.{hpp,cu}
public:
    RmsePostOutputLayer(cn_ctx *ctx, const json::Value &layerChild, Layer &precedingLayer);
    const std::string &type() const;
};
class BinaryClassificationLayer : public PostOutputLayer {    // layers/BinaryClassificationLayer.{hpp,cu}
public:
    BinaryClassificationLayer(cn_ctx *ctx, const json::Value &layerChild, Layer &precedingLayer);
    const std::string &type() const;
    int countCorrectClassifications();
};

}  // namespace layers
}  // namespace currennt_hip
