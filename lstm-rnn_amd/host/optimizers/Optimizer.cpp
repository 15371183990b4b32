#include "Optimizer.hpp"

#include <limits>

namespace currennt_hip {
namespace optimizers {

Optimizer::Optimizer(NeuralNetwork &neuralNetwork, data_sets::DataSet &trainingSet, data_sets::DataSet &validationSet,
                     data_sets::DataSet &testSet, int maxEpochs, int maxEpochsNoBest, int validateEvery, int testEvery,
                     bool hybridOnlineBatch)
    : m_neuralNetwork(neuralNetwork), m_trainingSet(trainingSet), m_validationSet(validationSet), m_testSet(testSet)
    , m_maxEpochs(maxEpochs), m_maxEpochsNoBest(maxEpochsNoBest), m_validateEvery(validateEvery), m_testEvery(testEvery)
    , m_hybridOnlineBatch(hybridOnlineBatch)
    , m_finished(false), m_curEpoch(0), m_epochsSinceLowestError(0)
    , m_lowestValidationError(std::numeric_limits<real_t>::max()), m_curTrainingError(std::numeric_limits<real_t>::max())
    , m_curValidationError(std::numeric_limits<real_t>::max()), m_curTestError(std::numeric_limits<real_t>::max())
    , m_curValidationClassError(0), m_curTrainingClassError(0), m_curTestClassError(0)
{
    m_bestWeights.resize(m_neuralNetwork.layers().size());
    m_curWeightUpdates.resize(m_neuralNetwork.layers().size());
    _storeWeights();
}

void Optimizer::_storeWeights()
{
    const std::vector<std::shared_ptr<layers::Layer> > &ls = m_neuralNetwork.layers();
    for (size_t i = 1; i + 1 < ls.size(); ++i) {
        layers::TrainableLayer *layer = dynamic_cast<layers::TrainableLayer *>(ls[i].get());
        if (layer) m_bestWeights[i] = layer->weights();
    }
}
void Optimizer::_restoreWeights()
{
    const std::vector<std::shared_ptr<layers::Layer> > &ls = m_neuralNetwork.layers();
    for (size_t i = 1; i + 1 < ls.size(); ++i) {
        layers::TrainableLayer *layer = dynamic_cast<layers::TrainableLayer *>(ls[i].get());
        if (layer) layer->setWeights(m_bestWeights[i]);
    }
}

real_t Optimizer::_processDataSet(data_sets::DataSet &ds, bool calcWeightUpdates, real_t *classError)
{
    real_t error = 0;
    *classError = (real_t)ds.totalTimesteps();
    const std::vector<std::shared_ptr<layers::Layer> > &ls = m_neuralNetwork.layers();
    const bool classification = dynamic_cast<layers::MulticlassClassificationLayer *>(&m_neuralNetwork.postOutputLayer()) != 0 ||
                                dynamic_cast<layers::BinaryClassificationLayer *>(&m_neuralNetwork.postOutputLayer()) != 0;   // Optimizer.cu:52-55

    data_sets::DataSetFraction frac;
    bool firstFraction = true;
    while (ds.getNextFraction(&frac)) {
        m_neuralNetwork.loadSequences(frac);
        m_neuralNetwork.computeForwardPass();
        float e = 0; int correct = 0;
        hipCheck(cn_loss_eval(m_neuralNetwork.postOutputLayer().handle(), &e, &correct), m_neuralNetwork.context());   // Optimizer.cu:46-55
        error += e;
        if (classification) *classError -= (real_t)correct;

        if (calcWeightUpdates) {
            m_neuralNetwork.computeBackwardPass();
            if (m_hybridOnlineBatch) {
                _updateWeights();                                           // Optimizer.cu:88-89
            } else {
                // batch learning: sum the fractions' weightUpdates, one update per epoch (:72-85, :95-97)
                for (size_t i = 1; i + 1 < ls.size(); ++i) {
                    layers::TrainableLayer *layer = dynamic_cast<layers::TrainableLayer *>(ls[i].get());
                    if (!layer) continue;
                    Hip::real_vector wu = layer->weightUpdates();
                    if (firstFraction) m_curWeightUpdates[i] = wu;
                    else for (size_t k = 0; k < wu.size(); ++k) m_curWeightUpdates[i][k] += wu[k];
                }
            }
        }
        firstFraction = false;
    }
    if (calcWeightUpdates && !m_hybridOnlineBatch) _updateWeights();
    error /= ds.totalSequences();                                           // :99-101
    *classError /= (real_t)ds.totalTimesteps();
    return error;
}

bool Optimizer::train()
{
    if (!m_finished) {
        ++m_curEpoch;
        m_curTrainingError = _processDataSet(m_trainingSet, true, &m_curTrainingClassError);
        if (!m_validationSet.empty() && m_curEpoch % m_validateEvery == 0) {
            m_curValidationError = _processDataSet(m_validationSet, false, &m_curValidationClassError);
            if (m_curValidationError < m_lowestValidationError) {
                m_lowestValidationError = m_curValidationError;
                m_epochsSinceLowestError = 0;
                _storeWeights();
            } else m_epochsSinceLowestError += m_validateEvery;
        } else if (m_validationSet.empty()) {
            m_epochsSinceLowestError = 0;
            _storeWeights();
        }
        if (!m_testSet.empty() && m_curEpoch % m_testEvery == 0)
            m_curTestError = _processDataSet(m_testSet, false, &m_curTestClassError);
        if (m_epochsSinceLowestError >= m_maxEpochsNoBest || (m_maxEpochs >= 0 && m_curEpoch >= m_maxEpochs)) {
            _restoreWeights();
            m_finished = true;
        }
    }
    return m_finished;
}

SteepestDescentOptimizer::SteepestDescentOptimizer(NeuralNetwork &neuralNetwork, data_sets::DataSet &trainingSet,
                                                   data_sets::DataSet &validationSet, data_sets::DataSet &testSet, int maxEpochs,
                                                   int maxEpochsNoBest, int validateEvery, int testEvery, real_t learningRate,
                                                   real_t momentum, bool hybridOnlineBatch)
    : Optimizer(neuralNetwork, trainingSet, validationSet, testSet, maxEpochs, maxEpochsNoBest, validateEvery, testEvery, hybridOnlineBatch)
    , m_learningRate(learningRate), m_momentum(momentum) {}

void SteepestDescentOptimizer::_updateWeights()
{
    const std::vector<std::shared_ptr<layers::Layer> > &ls = _neuralNetwork().layers();
    for (size_t i = 1; i + 1 < ls.size(); ++i) {
        layers::TrainableLayer *layer = dynamic_cast<layers::TrainableLayer *>(ls[i].get());
        if (!layer) continue;
        real_t lr = m_learningRate;
        if (layer->learningRate() >= 0.0) lr = layer->learningRate();        // SteepestDescentOptimizer.cu:78-80
        if (!hybridOnlineBatch()) {
            // batch mode: the epoch sum replaces the device weightUpdates before the update
            void *wu = cn_layer_device_ptr(layer->handle(), CN_BUF_WEIGHT_UPDATES);
            if (!wu) throw std::runtime_error("cannot address the weight updates of layer '" + layer->name() + "'");
            hipCheck(cn_layer_upload(layer->handle(), CN_BUF_WEIGHT_UPDATES, _curWeightUpdates()[i].data(), _curWeightUpdates()[i].size()),
                     _neuralNetwork().context());
        }
        hipCheck(cn_sgd_update(layer->handle(), lr, m_momentum), _neuralNetwork().context());
    }
}

}  // namespace optimizers
}  // namespace currennt_hip
