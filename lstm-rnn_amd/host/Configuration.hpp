// The handful of program options the hot path and its driver need (subset of
// currennt_lib/src/Configuration.cpp:120-190; same names, same defaults).  Options come from a
// `key = value` options file (positional argument or --options_file) and from `--key value` /
// `--key=value` on the command line; the command line wins (Configuration.cpp:192-217).
#pragma once

#include <map>
#include <string>
#include <vector>

#include "Types.hpp"

namespace currennt_hip {

class Configuration {
public:
    enum feedforwardformat_type_t { FORMAT_SINGLE_CSV, FORMAT_CSV, FORMAT_HTK };

    Configuration(int argc, const char *argv[]);

    bool trainingMode() const { return m_trainingMode; }
    bool hybridOnlineBatch() const { return m_hybridOnlineBatch; }
    bool shuffleFractions() const { return m_shuffleFractions; }
    bool shuffleSequences() const { return m_shuffleSequences; }
    bool listDevices() const { return m_listDevices; }
    bool revertStd() const { return m_revertStd; }
    int parallelSequences() const { return m_parallelSequences; }
    int maxEpochs() const { return m_maxEpochs; }
    int maxEpochsNoBest() const { return m_maxEpochsNoBest; }
    int validateEvery() const { return m_validateEvery; }
    int testEvery() const { return m_testEvery; }
    int truncateSeqLength() const { return m_truncSeqLength; }
    int outputFeatureKind() const { return m_outputFeatureKind; }
    unsigned randomSeed() const { return m_randomSeed; }
    real_t learningRate() const { return m_learningRate; }
    real_t momentum() const { return m_momentum; }
    real_t featurePeriod() const { return m_featurePeriod; }
    real_t trainingFraction() const { return m_trainingFraction; }
    real_t validationFraction() const { return m_validationFraction; }
    real_t testFraction() const { return m_testFraction; }
    real_t weightsDistributionUniformMin() const { return m_weightsUniformMin; }
    real_t weightsDistributionUniformMax() const { return m_weightsUniformMax; }
    real_t weightsDistributionNormalSigma() const { return m_weightsNormalSigma; }
    real_t weightsDistributionNormalMean() const { return m_weightsNormalMean; }
    bool weightsDistributionIsNormal() const { return m_weightsNormal; }
    feedforwardformat_type_t feedForwardFormat() const { return m_feedForwardFormat; }
    const std::string &networkFile() const { return m_networkFile; }
    const std::string &trainedNetworkFile() const { return m_trainedNetwork; }
    const std::string &feedForwardOutputFile() const { return m_feedForwardOutputFile; }
    const std::vector<std::string> &trainingFiles() const { return m_trainingFiles; }
    const std::vector<std::string> &validationFiles() const { return m_validationFiles; }
    const std::vector<std::string> &testFiles() const { return m_testFiles; }
    const std::vector<std::string> &feedForwardInputFiles() const { return m_feedForwardInputFiles; }
    cn_precision precision() const { return m_precision; }
    int device() const { return m_device; }
    // -1: the library's default (on for f32 / bf16x3, off for bf16); the library's "deterministic" option (cn_ctx_set_option)
    int deterministic() const { return m_deterministic; }
    // data-parallel training over `gpus` devices of this node, one process per GPU (no counterpart in the reference,
    // which drives one device: main.cpp:526-541); dpRank/dpWorld: shard of a host-only --dump_fractions run
    int gpus() const { return m_gpus; }
    int dpRank() const { return m_dpRank; }
    int dpWorld() const { return m_dpWorld; }
    bool help() const { return m_help; }
    bool dumpFractions() const { return m_dumpFractions; }
    int dumpEpochs() const { return m_dumpEpochs; }
    // data/noise options of Configuration.cpp:139-146,171-176
    real_t inputNoiseSigma() const { return m_inputNoiseSigma; }
    real_t weightNoiseSigma() const { return m_weightNoiseSigma; }
    int inputLeftContext() const { return m_inputLeftContext; }
    int inputRightContext() const { return m_inputRightContext; }
    int outputTimeLag() const { return m_outputTimeLag; }
    // autosave options of Configuration.cpp:159-164
    bool autosave() const { return m_autosave; }
    bool autosaveBest() const { return m_autosaveBest; }
    const std::string &autosavePrefix() const { return m_autosavePrefix; }
    const std::string &continueFile() const { return m_continueFile; }
    // "key=value;..." of every option given (the reference serialises its boost variables map, Configuration.cpp:44-67)
    const std::string &serializedOptions() const { return m_serializedOptions; }
    static const char *usage();

private:
    int m_dumpEpochs = 1;
    bool m_dumpFractions = false, m_autosave = false, m_autosaveBest = false;
    real_t m_inputNoiseSigma = 0, m_weightNoiseSigma = 0;
    int m_inputLeftContext = 0, m_inputRightContext = 0, m_outputTimeLag = 0;
    std::string m_autosavePrefix, m_continueFile, m_serializedOptions;
    bool m_help = false, m_trainingMode = false, m_hybridOnlineBatch = false, m_shuffleFractions = false,
         m_shuffleSequences = false, m_listDevices = false, m_revertStd = true, m_weightsNormal = false;
    int m_parallelSequences = 1, m_maxEpochs = -1, m_maxEpochsNoBest = 20, m_validateEvery = 1, m_testEvery = 1,
        m_truncSeqLength = 0, m_outputFeatureKind = 9, m_device = 0, m_gpus = 1, m_dpRank = 0, m_dpWorld = 1;
    unsigned m_randomSeed = 0;
    real_t m_learningRate = 1e-5f, m_momentum = 0.9f, m_featurePeriod = 10, m_trainingFraction = 1, m_validationFraction = 1,
           m_testFraction = 1, m_weightsUniformMin = -0.1f, m_weightsUniformMax = 0.1f, m_weightsNormalSigma = 0.1f, m_weightsNormalMean = 0;
    feedforwardformat_type_t m_feedForwardFormat = FORMAT_SINGLE_CSV;
    std::string m_networkFile = "network.jsn", m_trainedNetwork = "trained_network.jsn", m_feedForwardOutputFile = "ff_output.csv";
    std::vector<std::string> m_trainingFiles, m_validationFiles, m_testFiles, m_feedForwardInputFiles;
    cn_precision m_precision = CN_PREC_F32;
    int m_deterministic = -1;

    void apply(const std::string &key, const std::string &value);
};

}  // namespace currennt_hip
