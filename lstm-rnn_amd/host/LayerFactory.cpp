#include "LayerFactory.hpp"

#include <stdexcept>

namespace currennt_hip {

layers::Layer *LayerFactory::createLayer(cn_ctx *ctx, const std::string &layerType, const json::Value &layerChild,
                                         const json::Value *weightsSection, int parallelSequences, int maxSeqLength,
                                         layers::Layer *precedingLayer)
{
    using namespace layers;
    if (layerType == "input") return new InputLayer(ctx, layerChild, parallelSequences, maxSeqLength);
    if (!precedingLayer) throw std::runtime_error("Not an input layer and no preceding layer given");
    // type strings of LayerFactory.cu:52-87 that exist on the MI355X path
    if (layerType == "feedforward_tanh") return new FeedForwardLayer(ctx, layerChild, weightsSection, *precedingLayer, CN_LAYER_FF_TANH);
    if (layerType == "feedforward_logistic") return new FeedForwardLayer(ctx, layerChild, weightsSection, *precedingLayer, CN_LAYER_FF_LOGISTIC);
    if (layerType == "feedforward_identity") return new FeedForwardLayer(ctx, layerChild, weightsSection, *precedingLayer, CN_LAYER_FF_IDENTITY);
    if (layerType == "softmax") return new SoftmaxLayer(ctx, layerChild, weightsSection, *precedingLayer);
    if (layerType == "lstm") return new LstmLayer(ctx, layerChild, weightsSection, *precedingLayer, false);
    if (layerType == "blstm") return new LstmLayer(ctx, layerChild, weightsSection, *precedingLayer, true);
    if (layerType == "sse" || layerType == "weightedsse" || layerType == "rmse" || layerType == "ce" || layerType == "wf" ||
        layerType == "binary_classification" || layerType == "multiclass_classification") {
        if (!precedingLayer->isTrainable())                                                      // LayerFactory.cu:68-70
            throw std::runtime_error("Cannot add post output layer after a non trainable layer");
        if (layerType == "sse") return new SsePostOutputLayer(ctx, layerChild, *precedingLayer);
        if (layerType == "weightedsse") return new WeightedSsePostOutputLayer(ctx, layerChild, *precedingLayer);
        if (layerType == "rmse") return new RmsePostOutputLayer(ctx, layerChild, *precedingLayer);
        if (layerType == "ce") return new CePostOutputLayer(ctx, layerChild, *precedingLayer);
        if (layerType == "wf") return new SseMaskPostOutputLayer(ctx, layerChild, *precedingLayer);
        if (layerType == "binary_classification") return new BinaryClassificationLayer(ctx, layerChild, *precedingLayer);
        return new MulticlassClassificationLayer(ctx, layerChild, *precedingLayer);
    }
    throw std::runtime_error("Unknown layer type '" + layerType + "'");                          // LayerFactory.cu:86
}

}  // namespace currennt_hip
