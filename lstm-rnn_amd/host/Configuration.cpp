#include "Configuration.hpp"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace currennt_hip {

namespace {

std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
bool toBool(const std::string &key, const std::string &v)
{
    if (v == "true" || v == "1" || v == "on" || v == "yes") return true;
    if (v == "false" || v == "0" || v == "off" || v == "no") return false;
    throw std::runtime_error("Error while parsing the command line and/or options file: invalid value '" + v + "' for option '" + key + "'");
}
std::vector<std::string> splitList(const std::string &v)      // Configuration.cpp:219-226 (',' and ';' separated lists)
{
    std::vector<std::string> out;
    std::string cur;
    for (size_t i = 0; i <= v.size(); ++i) {
        if (i == v.size() || v[i] == ',' || v[i] == ';') { if (!trim(cur).empty()) out.push_back(trim(cur)); cur.clear(); }
        else cur += v[i];
    }
    return out;
}

}  // namespace

const char *Configuration::usage()
{
    return "Usage: currennt_hip [options] [options-file]\n"
           "  common:   --network F --parallel_sequences N --random_seed N --cuda B(ignored) --list_devices B\n"
           "            --precision f32|bf16|bf16x3 --device N\n"
           "            --deterministic B (gradient sums in a fixed order: bit-identical runs; default true for f32 / bf16x3)\n"
           "            --gpus N (training only: data-parallel over N GPUs of this node, devices --device .. --device+N-1;\n"
           "                      parallel_sequences is per GPU; gradients are summed with RCCL)\n"
           "  training: --train B --stochastic B (= --hybrid_online_batch) --shuffle_fractions B --shuffle_sequences B\n"
           "            --max_epochs N --max_epochs_no_best N --validate_every N --test_every N --learning_rate X\n"
           "            --momentum X --save_network F --train_file F[,F] --val_file F --test_file F --truncate_seq N\n"
           "            --train_fraction X --val_fraction X --test_fraction X\n"
           "            --weights_dist uniform|normal --weights_uniform_min X --weights_uniform_max X\n"
           "            --weights_normal_sigma X --weights_normal_mean X\n"
           "  forward:  --ff_input_file F --ff_output_file F --ff_output_format single_csv|csv|htk\n"
           "            --ff_output_kind N --feature_period X --revert_std B\n";
}

void Configuration::apply(const std::string &key, const std::string &v)
{
    if (key != "continue" && v.find_first_of(";\"\\") == std::string::npos) m_serializedOptions += key + "=" + v + ";";
    if (key == "network") m_networkFile = v;
    else if (key == "cuda") (void)toBool(key, v);                       // accepted for compatibility; this build always runs on the MI355X
    else if (key == "list_devices") m_listDevices = toBool(key, v);
    else if (key == "parallel_sequences") m_parallelSequences = atoi(v.c_str());
    else if (key == "random_seed") m_randomSeed = (unsigned)strtoul(v.c_str(), 0, 10);
    else if (key == "ff_output_format") {
        if (v == "single_csv") m_feedForwardFormat = FORMAT_SINGLE_CSV;
        else if (v == "csv") m_feedForwardFormat = FORMAT_CSV;
        else if (v == "htk") m_feedForwardFormat = FORMAT_HTK;
        else throw std::runtime_error("Error while parsing the command line and/or options file: unknown output format '" + v + "'");
    }
    else if (key == "ff_output_file") m_feedForwardOutputFile = v;
    else if (key == "ff_output_kind") m_outputFeatureKind = atoi(v.c_str());
    else if (key == "feature_period") m_featurePeriod = (real_t)atof(v.c_str());
    else if (key == "ff_input_file") m_feedForwardInputFiles = splitList(v);
    else if (key == "revert_std") m_revertStd = toBool(key, v);
    else if (key == "train") m_trainingMode = toBool(key, v);
    else if (key == "stochastic" || key == "hybrid_online_batch") m_hybridOnlineBatch = toBool(key, v);
    else if (key == "shuffle_fractions") m_shuffleFractions = toBool(key, v);
    else if (key == "shuffle_sequences") m_shuffleSequences = toBool(key, v);
    else if (key == "max_epochs") m_maxEpochs = atoi(v.c_str());
    else if (key == "max_epochs_no_best") m_maxEpochsNoBest = atoi(v.c_str());
    else if (key == "validate_every") m_validateEvery = atoi(v.c_str());
    else if (key == "test_every") m_testEvery = atoi(v.c_str());
    else if (key == "optimizer") { if (v != "steepest_descent") throw std::runtime_error("Error while parsing the command line and/or options file: unknown optimizer '" + v + "'"); }
    else if (key == "learning_rate") m_learningRate = (real_t)atof(v.c_str());
    else if (key == "momentum") m_momentum = (real_t)atof(v.c_str());
    else if (key == "save_network") m_trainedNetwork = v;
    else if (key == "train_file") m_trainingFiles = splitList(v);
    else if (key == "val_file") m_validationFiles = splitList(v);
    else if (key == "test_file") m_testFiles = splitList(v);
    else if (key == "train_fraction") m_trainingFraction = (real_t)atof(v.c_str());
    else if (key == "val_fraction") m_validationFraction = (real_t)atof(v.c_str());
    else if (key == "test_fraction") m_testFraction = (real_t)atof(v.c_str());
    else if (key == "truncate_seq") m_truncSeqLength = atoi(v.c_str());
    else if (key == "weights_dist") {
        if (v == "uniform") m_weightsNormal = false; else if (v == "normal") m_weightsNormal = true;
        else throw std::runtime_error("Error while parsing the command line and/or options file: unknown weights distribution '" + v + "'");
    }
    else if (key == "weights_uniform_min") m_weightsUniformMin = (real_t)atof(v.c_str());
    else if (key == "weights_uniform_max") m_weightsUniformMax = (real_t)atof(v.c_str());
    else if (key == "weights_normal_sigma") m_weightsNormalSigma = (real_t)atof(v.c_str());
    else if (key == "weights_normal_mean") m_weightsNormalMean = (real_t)atof(v.c_str());
    else if (key == "precision") {
        if (v == "f32" || v == "fp32") m_precision = CN_PREC_F32; else if (v == "bf16") m_precision = CN_PREC_BF16;
        else if (v == "bf16x3") m_precision = CN_PREC_BF16X3;
        else throw std::runtime_error("Error while parsing the command line and/or options file: unknown precision '" + v + "'");
    }
    else if (key == "deterministic") m_deterministic = toBool(key, v) ? 1 : 0;
    else if (key == "device") m_device = atoi(v.c_str());
    else if (key == "gpus") { m_gpus = atoi(v.c_str()); if (m_gpus < 1) throw std::runtime_error("Error while parsing the command line and/or options file: --gpus must be >= 1"); }
    else if (key == "dp_rank") m_dpRank = atoi(v.c_str());
    else if (key == "dp_world") m_dpWorld = atoi(v.c_str());
    else if (key == "dump_fractions") m_dumpFractions = toBool(key, v);
    else if (key == "dump_epochs") m_dumpEpochs = std::max(1, atoi(v.c_str()));
    else if (key == "input_noise_sigma") m_inputNoiseSigma = (real_t)atof(v.c_str());
    else if (key == "weight_noise_sigma") m_weightNoiseSigma = (real_t)atof(v.c_str());
    else if (key == "input_left_context") m_inputLeftContext = atoi(v.c_str());
    else if (key == "input_right_context") m_inputRightContext = atoi(v.c_str());
    else if (key == "output_time_lag") m_outputTimeLag = atoi(v.c_str());
    else if (key == "autosave") m_autosave = toBool(key, v);
    else if (key == "autosave_best") m_autosaveBest = toBool(key, v);
    else if (key == "autosave_prefix") m_autosavePrefix = v;
    else if (key == "continue") m_continueFile = v;
    else if (key == "cache_path") { /* accepted, unused: sequences are kept in RAM instead of a cache file */ }
    else throw std::runtime_error("Error while parsing the command line and/or options file: unknown option '" + key + "'");
}

Configuration::Configuration(int argc, const char *argv[])
{
    std::vector<std::pair<std::string, std::string> > cli;
    std::string optionsFile;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--help" || a == "-h") { m_help = true; continue; }
        if (a.compare(0, 2, "--") == 0) {
            std::string key = a.substr(2), val;
            size_t eq = key.find('=');
            if (eq != std::string::npos) { val = key.substr(eq + 1); key = key.substr(0, eq); }
            else if (i + 1 < argc) val = argv[++i];
            else throw std::runtime_error("Error while parsing the command line and/or options file: missing value for '" + key + "'");
            if (key == "options_file") optionsFile = val; else cli.push_back(std::make_pair(key, val));
        } else optionsFile = a;                                                  // positional options file, Configuration.cpp:192-194
    }
    if (!optionsFile.empty()) {
        std::ifstream f(optionsFile.c_str());
        if (!f.good()) throw std::runtime_error("Error while parsing the command line and/or options file: cannot open '" + optionsFile + "'");
        std::string line;
        while (std::getline(f, line)) {
            size_t hash = line.find('#');
            if (hash != std::string::npos) line = line.substr(0, hash);
            size_t eq = line.find('=');
            if (eq == std::string::npos) continue;
            apply(trim(line.substr(0, eq)), trim(line.substr(eq + 1)));
        }
    }
    // --continue: the options stored in the autosave file come first (Configuration.cpp:236-248)
    for (size_t i = 0; i < cli.size(); ++i)
        if (cli[i].first == "continue") {
            std::ifstream f(cli[i].second.c_str());
            if (!f.good()) throw std::runtime_error("Error while restoring configuration from autosave file: cannot open '" + cli[i].second + "'");
            std::stringstream ss; ss << f.rdbuf();
            const std::string text = ss.str(), tag = "\"configuration\"";
            size_t p = text.find(tag);
            if (p != std::string::npos && (p = text.find('"', text.find(':', p + tag.size()))) != std::string::npos) {
                size_t e = text.find('"', p + 1);
                std::vector<std::string> kv;
                std::string cur;
                for (size_t k = p + 1; k <= e && e != std::string::npos; ++k) {
                    if (k == e || text[k] == ';') { if (!cur.empty()) kv.push_back(cur); cur.clear(); } else cur += text[k];
                }
                for (size_t k = 0; k < kv.size(); ++k) {
                    size_t eq = kv[k].find('=');
                    if (eq != std::string::npos && kv[k].substr(0, eq) != "continue") apply(kv[k].substr(0, eq), kv[k].substr(eq + 1));
                }
            }
        }
    if (const char *dev = getenv("CURRENNT_CUDA_DEVICE")) m_device = atoi(dev);       // main.cpp:527-531 (kept under its old name)
    for (size_t i = 0; i < cli.size(); ++i) apply(cli[i].first, cli[i].second);   // command line wins
    if (m_parallelSequences < 1) throw std::runtime_error("Error while parsing the command line and/or options file: parallel_sequences must be >= 1");
}

}  // namespace currennt_hip
