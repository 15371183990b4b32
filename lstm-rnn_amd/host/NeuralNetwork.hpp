// NeuralNetwork (currennt_lib/src/NeuralNetwork.{hpp,cpp}): the layer stack built from a network JSON
// document, forward / backward iteration, error evaluation, export, getOutputs().
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "Json.hpp"
#include "layers/Layer.hpp"

namespace currennt_hip {

class NeuralNetwork {
public:
    // weightsInit: draws the initial weights of layers without a "weights" entry (TrainableLayer.cu:103-126)
    struct WeightsInit { bool normal; real_t uniformMin, uniformMax, normalSigma, normalMean; unsigned seed; };

    NeuralNetwork(const json::Value &jsonDoc, int parallelSequences, int maxSeqLength, int inputSizeOverride = -1,
                  cn_precision precision = CN_PREC_F32, int device = 0, const WeightsInit *weightsInit = 0);
    ~NeuralNetwork();

    const std::vector<std::shared_ptr<layers::Layer> > &layers() const { return m_layers; }
    layers::InputLayer &inputLayer();
    layers::TrainableLayer &outputLayer();
    layers::PostOutputLayer &postOutputLayer();

    void loadSequences(const data_sets::DataSetFraction &fraction);    // NeuralNetwork.cpp:161-166
    void prefetchSequences(const data_sets::DataSetFraction &fraction);   // cn_fraction_prefetch: what the next loadSequences will load
    void computeForwardPass();                                         // :168-173
    void computeBackwardPass();                                        // :175-184
    // Data-parallel training: bind this rank's RCCL communicator (collective; `id` = the CN_COMM_ID_BYTES rank 0 got from
    // cn_comm_unique_id).  Afterwards computeBackwardPass() all-reduces every trainable layer's weightUpdates right
    // behind that layer's backward pass (bucket = layer, beside the backward pass of the layers below).
    void initDataParallel(const char *id, int rank, int world);
    int dpWorld() const { return m_world; }
    int dpRank() const { return m_rank; }
    bool dataParallel() const { return m_dp; }
    // batch (non-stochastic) training sums the fractions' gradients on the host first and exchanges the epoch sum once
    // (Optimizer.cu:72-85,95-97): the per-fraction exchange is switched off for it
    void setExchangePerFraction(bool on) { m_exchangePerFraction = on; }
    real_t calculateError() const;                                     // :186-190
    void exportLayers(json::Value *jsonDoc) const;                     // :192-211
    void exportWeights(json::Value *jsonDoc) const;                    // :213-235
    std::vector<std::vector<std::vector<real_t> > > getOutputs();      // :237-262

    cn_ctx *context() const { return m_ctx; }

private:
    cn_ctx *m_ctx;
    int m_rank = 0, m_world = 1;
    bool m_dp = false, m_exchangePerFraction = true;
    std::vector<std::shared_ptr<layers::Layer> > m_layers;
};

}  // namespace currennt_hip
