#include "NeuralNetwork.hpp"

#include <random>
#include <stdexcept>

#include "LayerFactory.hpp"

namespace currennt_hip {

NeuralNetwork::NeuralNetwork(const json::Value &jsonDoc, int parallelSequences, int maxSeqLength, int inputSizeOverride,
                             cn_precision precision, int device, const WeightsInit *weightsInit)
    : m_ctx(0)
{
    hipCheck(cn_ctx_create(device, precision, 0, &m_ctx));
    try {
        try {
            if (!jsonDoc.isObject() || !jsonDoc.hasMember("layers")) throw std::runtime_error("Missing section 'layers'");
            const json::Value &layersSection = jsonDoc["layers"];
            if (!layersSection.isArray()) throw std::runtime_error("Section 'layers' is not an array");
            const json::Value *weightsSection = 0;
            if (jsonDoc.hasMember("weights")) {
                if (!jsonDoc["weights"].isObject()) throw std::runtime_error("Section 'weights' is not an object");
                weightsSection = &jsonDoc["weights"];
            }
            for (size_t i = 0; i < layersSection.size(); ++i) {
                json::Value layerChild = layersSection[i];
                if (!layerChild.isObject()) throw std::runtime_error("A layer section in the 'layers' array is not an object");
                if (!layerChild.hasMember("type")) throw std::runtime_error("Missing value 'type' in layer description");
                const std::string layerType = layerChild["type"].getString();
                if (inputSizeOverride > 0 && layerType == "input") layerChild["size"] = json::Value(inputSizeOverride);   // NeuralNetwork.cpp:71-73
                try {
                    layers::Layer *layer = LayerFactory::createLayer(m_ctx, layerType, layerChild, weightsSection, parallelSequences,
                                                                     maxSeqLength, m_layers.empty() ? 0 : m_layers.back().get());
                    m_layers.push_back(std::shared_ptr<layers::Layer>(layer));
                } catch (const std::exception &e) {
                    throw std::runtime_error(std::string("Could not create layer: ") + e.what());
                }
            }
            if (m_layers.size() < 3) throw std::runtime_error("Not enough layers defined");
            if (!dynamic_cast<layers::InputLayer *>(m_layers.front().get())) throw std::runtime_error("The first layer is not an input layer");
            for (size_t i = 1; i < m_layers.size(); ++i)
                if (dynamic_cast<layers::InputLayer *>(m_layers[i].get())) throw std::runtime_error("Multiple input layers defined");
            if (!m_layers.back()->isPostOutput()) throw std::runtime_error("The last layer is not a post output layer");
            for (size_t i = 0; i + 1 < m_layers.size(); ++i)
                if (m_layers[i]->isPostOutput()) throw std::runtime_error("Multiple post output layers defined");
            for (size_t i = 0; i < m_layers.size(); ++i)
                for (size_t j = 0; j < m_layers.size(); ++j)
                    if (i != j && m_layers[i]->name() == m_layers[j]->name())
                        throw std::runtime_error("Different layers have the same name '" + m_layers[i]->name() + "'");

            // Initial weights for layers that are not in the "weights" section: one generator shared by all layers in construction
            // order (TrainableLayer.cu:108-124: a function-local static boost::mt19937 seeded with --random_seed).  boost::mt19937
            // and std::mt19937 are the same engine (MT19937, 32-bit outputs); what differs between the libraries is how a
            // distribution turns engine outputs into floats, so the UNIFORM case (the default, Configuration.cpp:186-187) restates
            // Boost's published algorithm for boost::random::uniform_real_distribution<float>(0, range) -- generate_uniform_real:
            // result = float(engine()) / (float(engine.max()) + 1) * range + 0, drawn again while result >= range -- followed by
            // the reference's `+ uniformMin`.  Boost is not in this image, so this cannot be checked against a reference run here:
            // it is NOT part of any parity claim (every parity test passes explicit weights).  The normal case uses
            // std::normal_distribution (Boost's algorithm for it changed between releases).
            std::mt19937 gen(weightsInit && weightsInit->seed ? weightsInit->seed : 5489u);
            for (size_t i = 0; i < m_layers.size(); ++i) {
                layers::TrainableLayer *tl = dynamic_cast<layers::TrainableLayer *>(m_layers[i].get());
                if (!tl || (weightsSection && weightsSection->hasMember(tl->name()))) continue;
                Hip::real_vector w((size_t)tl->weightCount());
                if (weightsInit && weightsInit->normal) {
                    std::normal_distribution<real_t> dist(weightsInit->normalMean, weightsInit->normalSigma);
                    for (size_t k = 0; k < w.size(); ++k) w[k] = dist(gen);
                } else {
                    const real_t lo = weightsInit ? weightsInit->uniformMin : -0.1f, hi = weightsInit ? weightsInit->uniformMax : 0.1f;
                    const real_t range = hi - lo;
                    const real_t divisor = static_cast<real_t>(gen.max() - gen.min()) + 1;          // 2^32
                    for (size_t k = 0; k < w.size(); ++k) {
                        real_t r;
                        do { r = static_cast<real_t>(gen() - gen.min()) / divisor * range + 0; } while (range > 0 && !(r < range));
                        w[k] = r + lo;
                    }
                }
                tl->setWeights(w);
            }
        } catch (const std::exception &e) {
            throw std::runtime_error(std::string("Invalid network file: ") + e.what());
        }
    } catch (...) {
        m_layers.clear();
        cn_ctx_destroy(m_ctx);
        throw;
    }
}

NeuralNetwork::~NeuralNetwork()
{
    m_layers.clear();
    cn_ctx_destroy(m_ctx);
}

layers::InputLayer &NeuralNetwork::inputLayer() { return static_cast<layers::InputLayer &>(*m_layers.front()); }
layers::TrainableLayer &NeuralNetwork::outputLayer() { return static_cast<layers::TrainableLayer &>(*m_layers[m_layers.size() - 2]); }
layers::PostOutputLayer &NeuralNetwork::postOutputLayer() { return static_cast<layers::PostOutputLayer &>(*m_layers.back()); }

static cn_fraction describe(const data_sets::DataSetFraction &fraction)
{
    cn_fraction f;
    f.max_seq_length = fraction.maxSeqLength(); f.min_seq_length = fraction.minSeqLength();
    f.num_sequences = fraction.numSequences();
    f.input_pattern_size = fraction.inputPatternSize(); f.output_pattern_size = fraction.outputPatternSize();
    f.pat_types = fraction.patTypes().data(); f.inputs = fraction.inputs().data();
    f.target_classes = fraction.targetClasses().empty() ? 0 : fraction.targetClasses().data();
    f.targets = fraction.outputs().empty() ? 0 : fraction.outputs().data();
    return f;
}

void NeuralNetwork::prefetchSequences(const data_sets::DataSetFraction &fraction)
{
    const cn_fraction f = describe(fraction);
    hipCheck(cn_fraction_prefetch(m_ctx, m_layers.front()->handle(), m_layers.back()->handle(), &f), m_ctx);
}

void NeuralNetwork::loadSequences(const data_sets::DataSetFraction &fraction)
{
    for (size_t i = 0; i < m_layers.size(); ++i) m_layers[i]->loadSequences(fraction);   // shape checks + host-side bookkeeping
    const cn_fraction f = describe(fraction);
    // (no synchronisation: cn_fraction_load has copied the host vectors into pinned staging memory when it returns, so
    // the caller may release them, and the upload runs under the previous fraction's compute)
    hipCheck(cn_fraction_load(m_ctx, m_layers.front()->handle(), m_layers.back()->handle(), &f), m_ctx);
}

void NeuralNetwork::initDataParallel(const char *id, int rank, int world)
{
    hipCheck(cn_comm_init(m_ctx, id, rank, world), m_ctx);
    m_rank = rank; m_world = world; m_dp = true;
}

void NeuralNetwork::computeForwardPass()
{
    for (size_t i = 0; i < m_layers.size(); ++i) m_layers[i]->computeForwardPass();
}
void NeuralNetwork::computeBackwardPass()
{
    for (size_t i = m_layers.size(); i-- > 0;) {
        m_layers[i]->computeBackwardPass();
        if (m_dp && m_exchangePerFraction && dynamic_cast<layers::TrainableLayer *>(m_layers[i].get())) {
            cn_layer *h = m_layers[i]->handle();
            hipCheck(cn_allreduce_grads(m_ctx, &h, 1), m_ctx);
        }
    }
}
real_t NeuralNetwork::calculateError() const
{
    return static_cast<layers::PostOutputLayer &>(*m_layers.back()).calculateError();
}

void NeuralNetwork::exportLayers(json::Value *jsonDoc) const
{
    if (!jsonDoc->isObject()) throw std::runtime_error("JSON document root must be an object");
    json::Value layersArray(json::Value::Array);
    for (size_t i = 0; i < m_layers.size(); ++i) m_layers[i]->exportLayer(&layersArray);
    (*jsonDoc)["layers"] = layersArray;
}
void NeuralNetwork::exportWeights(json::Value *jsonDoc) const
{
    if (!jsonDoc->isObject()) throw std::runtime_error("JSON document root must be an object");
    json::Value weightsObject(json::Value::Object);
    for (size_t i = 0; i < m_layers.size(); ++i) {
        const layers::TrainableLayer *tl = dynamic_cast<const layers::TrainableLayer *>(m_layers[i].get());
        if (tl) tl->exportWeights(&weightsObject);
    }
    (*jsonDoc)["weights"] = weightsObject;
}

std::vector<std::vector<std::vector<real_t> > > NeuralNetwork::getOutputs()
{
    layers::TrainableLayer &ol = outputLayer();
    const Hip::real_vector out = ol.outputs();
    const Hip::pattype_vector &pat = ol.patTypes();
    std::vector<std::vector<std::vector<real_t> > > outputs;
    for (int patIdx = 0; patIdx < (int)pat.size(); ++patIdx) {
        switch (pat[patIdx]) {
        case PATTYPE_FIRST:
            outputs.resize(outputs.size() + 1);
            /* fall through */
        case PATTYPE_NORMAL:
        case PATTYPE_LAST: {
            const int psIdx = patIdx % ol.parallelSequences();
            outputs[psIdx].push_back(std::vector<real_t>(out.begin() + (size_t)patIdx * ol.size(), out.begin() + (size_t)(patIdx + 1) * ol.size()));
            break; }
        default: break;
        }
    }
    return outputs;
}

}  // namespace currennt_hip
