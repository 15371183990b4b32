// Epoch protocol and momentum SGD (currennt_lib/src/optimizers/{Optimizer,SteepestDescentOptimizer}.*).
// Gradient accumulation, the update and best-weight bookkeeping all stay on the device; the host only
// sequences them.
#pragma once

#include <random>
#include <vector>

#include "../Json.hpp"
#include "../NeuralNetwork.hpp"
#include "../data_sets/DataSet.hpp"

namespace currennt_hip {
namespace optimizers {

class Optimizer {
public:
    Optimizer(NeuralNetwork &neuralNetwork, data_sets::DataSet &trainingSet, data_sets::DataSet &validationSet,
              data_sets::DataSet &testSet, int maxEpochs, int maxEpochsNoBest, int validateEvery, int testEvery,
              bool hybridOnlineBatch);
    virtual ~Optimizer() {}

    // Gaussian weight noise during the backward pass (Optimizer.cu:58-68,82-84; --weight_noise_sigma)
    void setWeightNoise(real_t sigma, unsigned randomSeed) { m_weightNoiseSigma = sigma; m_noiseGen.seed(randomSeed); }
    // autosave / --continue (Optimizer.cu:326-358)
    virtual void exportState(json::Value *jsonDoc) const;
    virtual void importState(const json::Value &jsonDoc);

    bool finished() const { return m_finished; }
    int currentEpoch() const { return m_curEpoch; }
    real_t lowestValidationError() const { return m_lowestValidationError; }
    int epochsSinceLowestValidationError() const { return m_epochsSinceLowestError; }
    real_t curTrainingError() const { return m_curTrainingError; }
    real_t curValidationError() const { return m_curValidationError; }
    real_t curTestError() const { return m_curTestError; }
    real_t curTrainingClassError() const { return m_curTrainingClassError; }
    real_t curValidationClassError() const { return m_curValidationClassError; }
    real_t curTestClassError() const { return m_curTestClassError; }

    bool train();                                                       // Optimizer.cu:283-324

protected:
    virtual void _updateWeights() = 0;
    // hybrid online/batch learning without weight noise: the update of the coming backward pass may be applied layer by layer
    // behind each layer's gradient (cn_ctx_arm_update); _updateWeights() then only completes it
    virtual void _armUpdate() {}
    real_t _processDataSet(data_sets::DataSet &ds, bool calcWeightUpdates, real_t *classError);   // Optimizer.cu:37-104
    void _scoreValidationSet();                                         // validation part of train()
    bool _dueThisEpoch(const data_sets::DataSet &set, int every) const;
    bool _shouldStop() const;
    struct StateTable;                                                  // scalars of the autosave state (Optimizer.cpp)
    void _storeWeights();                                               // :151-158
    void _restoreWeights();                                             // :160-168
    NeuralNetwork &_neuralNetwork() { return m_neuralNetwork; }
    // (the reference's `_curWeightUpdates()`, the epoch sum of batch learning, lives on the device: cn_ctx_accumulate_updates /
    // cn_ctx_take_accumulated)

private:
    NeuralNetwork &m_neuralNetwork;
    data_sets::DataSet &m_trainingSet, &m_validationSet, &m_testSet;
    const int m_maxEpochs, m_maxEpochsNoBest, m_validateEvery, m_testEvery;
    const bool m_hybridOnlineBatch;
    bool m_finished;
    int m_curEpoch, m_epochsSinceLowestError;
    real_t m_lowestValidationError, m_curTrainingError, m_curValidationError, m_curTestError,
           m_curValidationClassError, m_curTrainingClassError, m_curTestClassError;
    std::vector<Hip::real_vector> m_bestWeights;
    real_t m_weightNoiseSigma = 0;
    std::mt19937 m_noiseGen;
protected:
    static void _exportWeights(json::Value *jsonDoc, const char *arrayName, const std::vector<Hip::real_vector> &weights);   // :106-123
    static void _importWeights(const json::Value &jsonDoc, const char *arrayName, std::vector<Hip::real_vector> *weights);   // :125-149
    bool hybridOnlineBatch() const { return m_hybridOnlineBatch; }
};

class SteepestDescentOptimizer : public Optimizer {
public:
    SteepestDescentOptimizer(NeuralNetwork &neuralNetwork, data_sets::DataSet &trainingSet, data_sets::DataSet &validationSet,
                             data_sets::DataSet &testSet, int maxEpochs, int maxEpochsNoBest, int validateEvery, int testEvery,
                             real_t learningRate, real_t momentum, bool hybridOnlineBatch);
    void exportState(json::Value *jsonDoc) const;                       // SteepestDescentOptimizer.cu:118-131
    void importState(const json::Value &jsonDoc);
protected:
    void _updateWeights();                                              // SteepestDescentOptimizer.cu:67-94
    void _armUpdate();
private:
    real_t m_learningRate, m_momentum;
};

}  // namespace optimizers
}  // namespace currennt_hip
