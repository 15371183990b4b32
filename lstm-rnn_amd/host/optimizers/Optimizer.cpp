#include "Optimizer.hpp"

#include <limits>
#include <vector>

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
    m_neuralNetwork.setExchangePerFraction(hybridOnlineBatch);
    m_bestWeights.resize(m_neuralNetwork.layers().size());
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

    // The reference reads the error of every fraction back (calculateError + countCorrectClassifications,
    // Optimizer.cu:46-55), which synchronises host and device once per fraction.  Here the per-fraction terms are
    // added up on the device, in the same order and in float like the reference's `error += ...`, and read back once
    // per pass over the data set: the host keeps enqueueing fractions while the device works.
    hipCheck(cn_loss_read(m_neuralNetwork.context(), nullptr, nullptr, 1), m_neuralNetwork.context());     // clear the sums
    data_sets::DataSetFraction frac, next;
    bool firstFraction = true;
    bool have = ds.getNextFraction(&frac);
    while (have) {
        m_neuralNetwork.loadSequences(frac);
        m_neuralNetwork.computeForwardPass();
        hipCheck(cn_loss_accumulate(m_neuralNetwork.postOutputLayer().handle()), m_neuralNetwork.context());
        // The data set's worker has the next fraction ready one ahead (DataSet.cpp:202-240,632-668); in training it goes on
        // across PCIe and through the re-layout beside this fraction's backward pass (cn_fraction_prefetch), and the load at
        // the top of the next iteration only exchanges buffers.  (`next` keeps its vectors' storage when it becomes `frac`:
        // the hint is matched by address.)
        const bool haveNext = ds.getNextFraction(&next);
        if (haveNext && calcWeightUpdates) m_neuralNetwork.prefetchSequences(next);

        if (calcWeightUpdates) {
            // weight noise: the forward pass above used the clean weights, the backward pass runs on noisy ones
            // and the clean weights come back before the update (Optimizer.cu:58-68, :82-84).  The noise is
            // drawn on the host like the reference's (TrainableLayer.cu:188-209), std::mt19937 instead of boost's.
            std::vector<Hip::real_vector> origWeights(ls.size());
            if (m_weightNoiseSigma > 0) {
                std::normal_distribution<real_t> dist(0.0f, m_weightNoiseSigma);
                for (size_t i = 1; i + 1 < ls.size(); ++i) {
                    layers::TrainableLayer *layer = dynamic_cast<layers::TrainableLayer *>(ls[i].get());
                    if (!layer) continue;
                    origWeights[i] = layer->weights();
                    Hip::real_vector noisy = origWeights[i];
                    for (size_t k = 0; k < noisy.size(); ++k) noisy[k] += dist(m_noiseGen);
                    layer->setWeights(noisy);
                }
            }
            if (m_hybridOnlineBatch && !(m_weightNoiseSigma > 0)) _armUpdate();
            m_neuralNetwork.computeBackwardPass();
            if (m_weightNoiseSigma > 0) {
                hipCheck(cn_ctx_join(m_neuralNetwork.context()), m_neuralNetwork.context());   // gradient GEMMs still read the noisy operands
                for (size_t i = 1; i + 1 < ls.size(); ++i) {
                    layers::TrainableLayer *layer = dynamic_cast<layers::TrainableLayer *>(ls[i].get());
                    if (layer) layer->setWeights(origWeights[i]);
                }
            }
            if (m_hybridOnlineBatch) {
                _updateWeights();                                           // Optimizer.cu:88-89
            } else {
                // batch learning: sum the fractions' weightUpdates, one update per epoch (:72-85, :95-97) -- on the device,
                // one launch over all layers, like the reference's thrust::copy / thrust::transform(plus)
                hipCheck(cn_ctx_accumulate_updates(m_neuralNetwork.context(), firstFraction ? 1 : 0), m_neuralNetwork.context());
            }
        }
        firstFraction = false;
        have = haveNext;
        if (have) std::swap(frac, next);
    }
    if (calcWeightUpdates && !m_hybridOnlineBatch) _updateWeights();
    {
        float e = 0; int64_t correct = 0;
        // data-parallel: the sums of all ranks (every rank then computes the same epoch errors and takes the same
        // early-stopping decisions)
        if (m_neuralNetwork.dataParallel())
            hipCheck(cn_loss_read_global(m_neuralNetwork.context(), &e, &correct, 1), m_neuralNetwork.context());
        else
            hipCheck(cn_loss_read(m_neuralNetwork.context(), &e, &correct, 1), m_neuralNetwork.context());
        error = e;
        if (classification) *classError -= (real_t)correct;
    }
    error /= ds.totalSequences();                                           // :99-101
    *classError /= (real_t)ds.totalTimesteps();
    return error;
}

// One epoch (Optimizer.cu:283-324): train, then -- on the epochs the options ask for -- score the validation and test sets,
// keep the weights of the best validation epoch, and decide whether to stop.  Split into the three questions an epoch asks.
bool Optimizer::train()
{
    if (m_finished) return true;
    ++m_curEpoch;
    m_curTrainingError = _processDataSet(m_trainingSet, true, &m_curTrainingClassError);
    _scoreValidationSet();
    if (_dueThisEpoch(m_testSet, m_testEvery)) m_curTestError = _processDataSet(m_testSet, false, &m_curTestClassError);
    if (_shouldStop()) {
        _restoreWeights();          // training ends on the best weights seen, not on the last ones
        m_finished = true;
    }
    return m_finished;
}

bool Optimizer::_dueThisEpoch(const data_sets::DataSet &set, int every) const
{
    return !set.empty() && m_curEpoch % every == 0;
}

void Optimizer::_scoreValidationSet()
{
    if (m_validationSet.empty()) {
        // nothing to select on: every epoch counts as the best so far
        m_epochsSinceLowestError = 0;
        _storeWeights();
        return;
    }
    if (!_dueThisEpoch(m_validationSet, m_validateEvery)) return;
    m_curValidationError = _processDataSet(m_validationSet, false, &m_curValidationClassError);
    const bool improved = m_curValidationError < m_lowestValidationError;
    if (improved) {
        m_lowestValidationError = m_curValidationError;
        m_epochsSinceLowestError = 0;
        _storeWeights();
    } else {
        m_epochsSinceLowestError += m_validateEvery;
    }
}

bool Optimizer::_shouldStop() const
{
    const bool noProgress = m_epochsSinceLowestError >= m_maxEpochsNoBest;
    const bool outOfEpochs = m_maxEpochs >= 0 && m_curEpoch >= m_maxEpochs;
    return noProgress || outOfEpochs;
}

void Optimizer::_exportWeights(json::Value *jsonDoc, const char *arrayName, const std::vector<Hip::real_vector> &weights)
{
    json::Value weightsArray(json::Value::Array);
    for (size_t i = 0; i < weights.size(); ++i) {
        json::Value v(json::Value::Array);
        v.reserve(weights[i].size());
        for (size_t j = 0; j < weights[i].size(); ++j) v.pushBack(json::Value((double)weights[i][j]));
        weightsArray.pushBack(v);
    }
    jsonDoc->addMember(arrayName, weightsArray);
}
void Optimizer::_importWeights(const json::Value &jsonDoc, const char *arrayName, std::vector<Hip::real_vector> *weights)
{
    if (!jsonDoc.hasMember(arrayName) || !jsonDoc[arrayName].isArray())
        throw std::runtime_error(std::string("Array '") + arrayName + "' is missing or has the wrong type");
    const json::Value &arr = jsonDoc[arrayName];
    if (arr.size() != weights->size()) throw std::runtime_error(std::string("Array '") + arrayName + "' has a wrong size");
    for (size_t i = 0; i < arr.size(); ++i) {
        if (!arr[i].isArray()) throw std::runtime_error(std::string("Object in '") + arrayName + "' is not an array");
        if (arr[i].size() != (*weights)[i].size()) throw std::runtime_error(std::string("Subarray in '") + arrayName + "' has a wrong size");
        for (size_t j = 0; j < arr[i].size(); ++j) (*weights)[i][j] = (real_t)arr[i][j].getDouble();
    }
}

// Autosave state (Optimizer.cu:326-358; the JSON keys are the reference's file format).  One table names every scalar of
// the state once; export and import both walk it.
namespace {
struct IntField  { const char *key; int Optimizer::*member; };
struct RealField { const char *key; real_t Optimizer::*member; };
}  // namespace

struct Optimizer::StateTable {
    static const std::vector<IntField> &ints()
    {
        static const std::vector<IntField> t = {
            {"optimizer_cur_epoch", &Optimizer::m_curEpoch},
            {"optimizer_epochs_since_lowest_error", &Optimizer::m_epochsSinceLowestError},
        };
        return t;
    }
    static const std::vector<RealField> &reals()
    {
        static const std::vector<RealField> t = {
            {"optimizer_lowest_validation_error", &Optimizer::m_lowestValidationError},
            {"optimizer_cur_training_error", &Optimizer::m_curTrainingError},
            {"optimizer_cur_validation_error", &Optimizer::m_curValidationError},
            {"optimizer_cur_test_error", &Optimizer::m_curTestError},
            {"optimizer_cur_training_class_error", &Optimizer::m_curTrainingClassError},
            {"optimizer_cur_validation_class_error", &Optimizer::m_curValidationClassError},
            {"optimizer_cur_test_class_error", &Optimizer::m_curTestClassError},
        };
        return t;
    }
};

void Optimizer::exportState(json::Value *jsonDoc) const
{
    jsonDoc->addMember("optimizer_finished", json::Value(m_finished));
    for (const IntField &f : StateTable::ints()) jsonDoc->addMember(f.key, json::Value(this->*f.member));
    for (const RealField &f : StateTable::reals()) jsonDoc->addMember(f.key, json::Value((double)(this->*f.member)));
    _exportWeights(jsonDoc, "optimizer_best_weights", m_bestWeights);
}
void Optimizer::importState(const json::Value &jsonDoc)
{
    m_finished = jsonDoc["optimizer_finished"].getBool();
    for (const IntField &f : StateTable::ints()) this->*f.member = jsonDoc[f.key].getInt();
    for (const RealField &f : StateTable::reals()) this->*f.member = (real_t)jsonDoc[f.key].getDouble();
    _importWeights(jsonDoc, "optimizer_best_weights", &m_bestWeights);
}

// the momentum state lives on the device (CN_BUF_WEIGHT_DELTAS); autosave moves it through the host
void SteepestDescentOptimizer::exportState(json::Value *jsonDoc) const
{
    Optimizer::exportState(jsonDoc);
    NeuralNetwork &nn = const_cast<SteepestDescentOptimizer *>(this)->_neuralNetwork();
    const std::vector<std::shared_ptr<layers::Layer> > &ls = nn.layers();
    std::vector<Hip::real_vector> deltas(ls.size());
    for (size_t i = 1; i + 1 < ls.size(); ++i) {
        layers::TrainableLayer *layer = dynamic_cast<layers::TrainableLayer *>(ls[i].get());
        if (!layer) continue;
        deltas[i].resize(layer->weightCount());
        hipCheck(cn_layer_read(layer->handle(), CN_BUF_WEIGHT_DELTAS, 0, deltas[i].data(), deltas[i].size()), nn.context());
    }
    _exportWeights(jsonDoc, "steepest_descent_optimizer_weight_deltas", deltas);
}
void SteepestDescentOptimizer::importState(const json::Value &jsonDoc)
{
    Optimizer::importState(jsonDoc);
    NeuralNetwork &nn = _neuralNetwork();
    const std::vector<std::shared_ptr<layers::Layer> > &ls = nn.layers();
    std::vector<Hip::real_vector> deltas(ls.size());
    for (size_t i = 1; i + 1 < ls.size(); ++i) {
        layers::TrainableLayer *layer = dynamic_cast<layers::TrainableLayer *>(ls[i].get());
        if (layer) deltas[i].resize(layer->weightCount());
    }
    _importWeights(jsonDoc, "steepest_descent_optimizer_weight_deltas", &deltas);
    for (size_t i = 1; i + 1 < ls.size(); ++i) {
        layers::TrainableLayer *layer = dynamic_cast<layers::TrainableLayer *>(ls[i].get());
        if (layer) hipCheck(cn_layer_upload(layer->handle(), CN_BUF_WEIGHT_DELTAS, deltas[i].data(), deltas[i].size()), nn.context());
    }
}

SteepestDescentOptimizer::SteepestDescentOptimizer(NeuralNetwork &neuralNetwork, data_sets::DataSet &trainingSet,
                                                   data_sets::DataSet &validationSet, data_sets::DataSet &testSet, int maxEpochs,
                                                   int maxEpochsNoBest, int validateEvery, int testEvery, real_t learningRate,
                                                   real_t momentum, bool hybridOnlineBatch)
    : Optimizer(neuralNetwork, trainingSet, validationSet, testSet, maxEpochs, maxEpochsNoBest, validateEvery, testEvery, hybridOnlineBatch)
    , m_learningRate(learningRate), m_momentum(momentum) {}

void SteepestDescentOptimizer::_armUpdate()
{
    NeuralNetwork &nn = _neuralNetwork();
    hipCheck(cn_ctx_arm_update(nn.context(), m_learningRate, m_momentum), nn.context());
}

void SteepestDescentOptimizer::_updateWeights()
{
    NeuralNetwork &nn = _neuralNetwork();
    const std::vector<std::shared_ptr<layers::Layer> > &ls = nn.layers();
    if (!hybridOnlineBatch()) {
        // batch mode: the epoch sum becomes the weightUpdates again before the update (device to device); data-parallel ranks
        // then add their epoch sums up in one exchange over the whole arena
        hipCheck(cn_ctx_take_accumulated(nn.context()), nn.context());
        if (nn.dataParallel()) hipCheck(cn_allreduce_grads(nn.context(), 0, 0), nn.context());
    }
    for (size_t i = 1; i + 1 < ls.size(); ++i) {
        layers::TrainableLayer *layer = dynamic_cast<layers::TrainableLayer *>(ls[i].get());
        if (!layer) continue;
        real_t lr = m_learningRate;
        if (layer->learningRate() >= 0.0) lr = layer->learningRate();        // SteepestDescentOptimizer.cu:78-80
        hipCheck(cn_sgd_update(layer->handle(), lr, m_momentum), nn.context());
    }
}

}  // namespace optimizers
}  // namespace currennt_hip
