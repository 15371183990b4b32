// LayerFactory::createLayer (currennt_lib/src/LayerFactory.{hpp,cu}): type string -> layer object.
#pragma once

#include <string>

#include "layers/Layer.hpp"

namespace currennt_hip {

class LayerFactory {
public:
    static layers::Layer *createLayer(cn_ctx *ctx, const std::string &layerType, const json::Value &layerChild,
                                      const json::Value *weightsSection, int parallelSequences, int maxSeqLength,
                                      layers::Layer *precedingLayer = 0);
};

}  // namespace currennt_hip
