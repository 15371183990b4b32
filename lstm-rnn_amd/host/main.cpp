// currennt_hip: a `currennt`-compatible driver over the MI355X library (subset of currennt/src/main.cpp:
// training loop with the progress table, trained_network.jsn export, forward pass writers
// single_csv / csv / htk).  Exit code 2 and "FAILED: msg" on any error like main.cpp:492-495.
#include <signal.h>
#include <sys/stat.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>

#include "Configuration.hpp"
#include "NeuralNetwork.hpp"
#include "data_sets/DataSet.hpp"
#include "optimizers/Optimizer.hpp"

using namespace currennt_hip;

namespace {

enum data_set_type { DATA_SET_TRAINING, DATA_SET_VALIDATION, DATA_SET_TEST, DATA_SET_FEEDFORWARD };

// The driver's test hooks (CN_DP_FORCE, CN_DP_SAME_DEVICE, CN_DP_TEST_FAIL_RANK) only exist under the master switch
// CN_TEST_HOOKS=1: a compute job that inherits one of those variables by accident is not affected.
const char *testHook(const char *name)
{
    static const bool enabled = [] { const char *e = getenv("CN_TEST_HOOKS"); return e && atoi(e) != 0; }();
    return enabled ? getenv(name) : nullptr;
}

// This process' place in a data-parallel run (--gpus N: one process per GPU, forked by main() before any GPU call).
// idPipe: rank 0 writes the RCCL rendezvous id to every other rank's pipe, rank r > 0 reads its own.
struct DataParallel {
    int rank = 0, world = 1;
    bool active = false;                       // go through the communicator even with world == 1 (CN_DP_FORCE=1: test hook)
    std::vector<int> idWriteFds; int idReadFd = -1;
};

std::shared_ptr<data_sets::DataSet> loadDataSet(const Configuration &config, data_set_type dsType, const DataParallel &dp = DataParallel())   // main.cpp:574-640
{
    std::string type; std::vector<std::string> filenames; real_t fraction = 1; bool fracShuf = false, seqShuf = false; int truncSeqLength = 0;
    data_sets::DataSet::Augment augment;            // context and lag apply to every set (DataSet.cpp:302-305), noise to two (main.cpp:593,613)
    augment.contextLeft = config.inputLeftContext(); augment.contextRight = config.inputRightContext(); augment.outputLag = config.outputTimeLag();
    switch (dsType) {
    case DATA_SET_TRAINING: type = "training set"; filenames = config.trainingFiles(); fraction = config.trainingFraction();
        fracShuf = config.shuffleFractions(); seqShuf = config.shuffleSequences(); truncSeqLength = config.truncateSeqLength();
        augment.noiseDeviation = config.inputNoiseSigma(); break;
    case DATA_SET_VALIDATION: type = "validation set"; filenames = config.validationFiles(); fraction = config.validationFraction(); break;
    case DATA_SET_TEST: type = "test set"; filenames = config.testFiles(); fraction = config.testFraction(); break;
    default: type = "feed forward input set"; filenames = config.feedForwardInputFiles(); augment.noiseDeviation = config.inputNoiseSigma(); break;
    }
    printf("Loading %s ", type.c_str());
    for (size_t i = 0; i < filenames.size(); ++i) printf("'%s' ", filenames[i].c_str());
    printf("...");
    fflush(stdout);
    if (filenames.empty()) throw std::runtime_error("No " + type + " file given");
    std::shared_ptr<data_sets::DataSet> ds = std::make_shared<data_sets::DataSet>(
        filenames, config.parallelSequences(), fraction, truncSeqLength, fracShuf, seqShuf, config.trainingMode(), config.randomSeed(), augment);
    if (dp.world > 1) ds->setShard(dp.rank, dp.world);
    else if (config.dpWorld() > 1) ds->setShard(config.dpRank(), config.dpWorld());            // host-only --dump_fractions of one shard
    printf("done.\n");
    printf("Loaded fraction:  %d%%\n", (int)(fraction * 100));
    printf("Sequences:        %d\n", ds->totalSequences());
    printf("Sequence lengths: %d..%d\n", ds->minSeqLength(), ds->maxSeqLength());
    printf("Total timesteps:  %d\n\n", ds->totalTimesteps());
    return ds;
}

void printLayers(const NeuralNetwork &nn)                                                       // main.cpp:643-657
{
    int weights = 0;
    for (size_t i = 0; i < nn.layers().size(); ++i) {
        printf("(%d) %s ", (int)i, nn.layers()[i]->type().c_str());
        printf("[size: %d", nn.layers()[i]->size());
        const layers::TrainableLayer *tl = dynamic_cast<const layers::TrainableLayer *>(nn.layers()[i].get());
        if (tl) { printf(", bias: %.1lf, weights: %d", (double)tl->bias(), tl->weightCount()); weights += tl->weightCount(); }
        printf("]\n");
    }
    printf("Total weights: %d\n", weights);
}

void saveNetwork(const NeuralNetwork &nn, const std::string &filename)                          // main.cpp:681-698
{
    json::Value doc(json::Value::Object);
    nn.exportLayers(&doc);
    nn.exportWeights(&doc);
    doc.writeFile(filename);
}

std::string replaceAll(std::string s, const std::string &from, const std::string &to)
{
    for (size_t p = 0; (p = s.find(from, p)) != std::string::npos; p += to.size()) s.replace(p, from.size(), to);
    return s;
}

// "<prefix>_epochNNN.autosave": options, progress rows, network, weights and optimizer state (main.cpp:701-742)
void saveState(const Configuration &config, const NeuralNetwork &nn, const optimizers::Optimizer &optimizer, const std::string &infoRows)
{
    json::Value doc(json::Value::Object);
    doc.addMember("configuration", json::Value(config.serializedOptions()));
    doc.addMember("info_rows", json::Value(replaceAll(infoRows, "\n", ";;;")));
    nn.exportLayers(&doc);
    nn.exportWeights(&doc);
    optimizer.exportState(&doc);
    char epoch[32]; snprintf(epoch, sizeof(epoch), "epoch%03d.autosave", optimizer.currentEpoch());
    const std::string &prefix = config.autosavePrefix();
    doc.writeFile(prefix + (prefix.empty() ? "" : "_") + epoch);
}

void restoreState(const json::Value &doc, optimizers::Optimizer *optimizer, std::string *infoRows)   // main.cpp:744-758
{
    if (!doc.hasMember("info_rows")) throw std::runtime_error("Missing value 'info_rows'");
    *infoRows = replaceAll(doc["info_rows"].getString(), ";;;", "\n");
    optimizer->importState(doc);
}

std::string printfRow(const char *format, ...)                                                  // main.cpp:760-775
{
    char buffer[100];
    va_list args; va_start(args, format); vsnprintf(buffer, sizeof(buffer), format, args); va_end(args);
    std::cout << buffer; fflush(stdout);
    return std::string(buffer);
}

void makeDirs(const std::string &path)
{
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') mkdir(path.substr(0, i).c_str(), 0777);
}
void swap32(void *p) { unsigned char *b = (unsigned char *)p; std::swap(b[0], b[3]); std::swap(b[1], b[2]); }
void swap16(void *p) { unsigned char *b = (unsigned char *)p; std::swap(b[0], b[1]); }

void feedForward(const Configuration &config, NeuralNetwork &nn, data_sets::DataSet &set)      // main.cpp:307-490
{
    const Hip::real_vector means = set.outputMeans(), stdevs = set.outputStdevs();
    const bool unstandardize = config.revertStd();
    if (unstandardize) printf("Outputs will be scaled by mean and standard deviation specified in NC file.\n");
    std::ofstream single;
    if (config.feedForwardFormat() == Configuration::FORMAT_SINGLE_CSV) single.open(config.feedForwardOutputFile().c_str());
    data_sets::DataSetFraction frac;
    int fracIdx = 0;
    while (set.getNextFraction(&frac)) {
        printf("Computing outputs for data fraction %d...", ++fracIdx);
        fflush(stdout);
        nn.loadSequences(frac);
        nn.computeForwardPass();
        std::vector<std::vector<std::vector<real_t> > > outputs = nn.getOutputs();
        for (int psIdx = 0; psIdx < (int)outputs.size(); ++psIdx) {
            const std::string &tag = frac.seqInfo(psIdx).seqTag;
            auto value = [&](int t, int o) { real_t v = outputs[psIdx][t][o]; if (unstandardize) { v *= stdevs[o]; v += means[o]; } return v; };
            if (config.feedForwardFormat() == Configuration::FORMAT_SINGLE_CSV) {
                single << tag;
                for (size_t t = 0; t < outputs[psIdx].size(); ++t)
                    for (size_t o = 0; o < outputs[psIdx][t].size(); ++o) single << ';' << value((int)t, (int)o);
                single << '\n';
                continue;
            }
            // one file per sequence below <ff_output_file>/<dir of tag>/
            std::string rel = tag;
            while (!rel.empty() && rel[0] == '/') rel.erase(0, 1);
            size_t slash = rel.find_last_of('/');
            std::string dir = config.feedForwardOutputFile() + (slash == std::string::npos ? "" : "/" + rel.substr(0, slash));
            std::string base = slash == std::string::npos ? rel : rel.substr(slash + 1);
            makeDirs(dir);
            if (config.feedForwardFormat() == Configuration::FORMAT_CSV) {
                size_t dot = base.find_last_of('.');
                if (dot != std::string::npos && dot > 0) base = base.substr(0, dot);
                std::ofstream file((dir + "/" + base + ".csv").c_str());
                for (size_t t = 0; t < outputs[psIdx].size(); ++t) {
                    for (size_t o = 0; o < outputs[psIdx][t].size(); ++o) { if (o) file << ';'; file << value((int)t, (int)o); }
                    file << '\n';
                }
            } else if (!outputs[psIdx].empty()) {                                                // FORMAT_HTK, main.cpp:432-480
                std::ofstream file((dir + "/" + base + ".htk").c_str(), std::ofstream::out | std::ios::binary);
                unsigned tmp = (unsigned)outputs[psIdx].size(); swap32(&tmp); file.write((const char *)&tmp, 4);
                tmp = (unsigned)(config.featurePeriod() * 1e4); swap32(&tmp); file.write((const char *)&tmp, 4);
                unsigned short tmp2 = (unsigned short)(outputs[psIdx][0].size() * sizeof(float)); swap16(&tmp2); file.write((const char *)&tmp2, 2);
                tmp2 = (unsigned short)config.outputFeatureKind(); swap16(&tmp2); file.write((const char *)&tmp2, 2);
                for (size_t t = 0; t < outputs[psIdx].size(); ++t)
                    for (size_t o = 0; o < outputs[psIdx][t].size(); ++o) { float v = value((int)t, (int)o); swap32(&v); file.write((const char *)&v, 4); }
            }
        }
        printf(" done.\n");
    }
}

int trainerMain(const Configuration &config, const DataParallel &dp = DataParallel())           // main.cpp:97-498
{
    const bool root = dp.rank == 0;            // only rank 0 writes files (all ranks hold identical weights)
    try {
        const std::string networkFile = config.continueFile().empty() ? config.networkFile() : config.continueFile();   // main.cpp:102
        printf("Reading network from '%s'... ", networkFile.c_str());
        fflush(stdout);
        json::Value netDoc = json::Value::parseFile(networkFile);
        printf("done.\n\n");

        std::shared_ptr<data_sets::DataSet> trainingSet = std::make_shared<data_sets::DataSet>(), validationSet = trainingSet,
                                             testSet = trainingSet, feedForwardSet = trainingSet;
        validationSet = std::make_shared<data_sets::DataSet>(); testSet = std::make_shared<data_sets::DataSet>();
        feedForwardSet = std::make_shared<data_sets::DataSet>();
        if (config.trainingMode()) {
            trainingSet = loadDataSet(config, DATA_SET_TRAINING, dp);
            if (!config.validationFiles().empty()) validationSet = loadDataSet(config, DATA_SET_VALIDATION, dp);
            if (!config.testFiles().empty()) testSet = loadDataSet(config, DATA_SET_TEST, dp);
        } else feedForwardSet = loadDataSet(config, DATA_SET_FEEDFORWARD);

        int maxSeqLength = config.trainingMode()
            ? std::max(trainingSet->maxSeqLength(), std::max(validationSet->maxSeqLength(), testSet->maxSeqLength()))
            : feedForwardSet->maxSeqLength();

        if (config.dumpFractions()) {     // host-only check of reader + packer (no GPU needed)
            data_sets::DataSet &ds = config.trainingMode() ? *trainingSet : *feedForwardSet;
            data_sets::DataSetFraction frac;
            // --dump_epochs N: N passes over the set (the shuffles of --shuffle_sequences / --shuffle_fractions run at the start
            // of every pass, DataSet.cpp:416-427)
            for (int epoch = 0; epoch < config.dumpEpochs(); ++epoch) {
                if (config.dumpEpochs() > 1) printf("EPOCH %d\n", epoch);
                int idx = 0;
                while (ds.getNextFraction(&frac)) {
                    double sx = 0; long st = 0; int none = 0;
                    for (size_t i = 0; i < frac.inputs().size(); ++i) sx += frac.inputs()[i];
                    for (size_t i = 0; i < frac.targetClasses().size(); ++i) if (frac.targetClasses()[i] >= 0) st += frac.targetClasses()[i];
                    for (size_t i = 0; i < frac.outputs().size(); ++i) sx += 1000.0 * frac.outputs()[i];
                    for (size_t i = 0; i < frac.patTypes().size(); ++i) none += frac.patTypes()[i] == PATTYPE_NONE;
                    // tags / lens / pieces = the sequences of the fraction, slot by slot: tag, length and piece number under
                    // --truncate_seq (the length sort is std::sort like the reference's, DataSet.cpp:603-605: ties land in an
                    // implementation-defined order, which a checker has to be told)
                    std::string order, lens, pieces;
                    for (int i = 0; i < frac.numSequences(); ++i) {
                        order += (i ? "," : "") + frac.seqInfo(i).seqTag;
                        lens += (i ? "," : "") + std::to_string(frac.seqInfo(i).length);
                        pieces += (i ? "," : "") + std::to_string(frac.seqInfo(i).originalSeqIdx);
                    }
                    const bool any = frac.numSequences() > 0;
                    printf("FRACTION %d T=%d Tmin=%d seqs=%d none=%d sum_inputs=%.6f sum_targets=%ld first_tag=%s tags=%s lens=%s pieces=%s\n", idx++,
                           frac.maxSeqLength(), frac.minSeqLength(), frac.numSequences(), none, sx, st, any ? frac.seqInfo(0).seqTag.c_str() : "-",
                           any ? order.c_str() : "-", any ? lens.c_str() : "-", any ? pieces.c_str() : "-");
                }
            }
            return 0;
        }

        printf("Creating the neural network... ");
        fflush(stdout);
        // main.cpp:146-148 overrides the input layer size with the training set's pattern size; with context
        // splicing the fractions carry (left + right + 1) frames per pattern, so that is the size used here
        // (the reference passes the unspliced size and then fails in InputLayer::loadSequences).
        const int inputSize = config.trainingMode() ? trainingSet->fractionInputPatternSize() : -1;
        NeuralNetwork::WeightsInit wi = { config.weightsDistributionIsNormal(), config.weightsDistributionUniformMin(),
                                          config.weightsDistributionUniformMax(), config.weightsDistributionNormalSigma(),
                                          config.weightsDistributionNormalMean(), config.randomSeed() };
        // one rank per GPU: device --device + rank.  CN_DP_SAME_DEVICE=1 (tests on a one-GPU box, together with the library's
        // CN_COMM_BACKEND=ipc): every rank on --device
        const bool sameDevice = dp.active && testHook("CN_DP_SAME_DEVICE") != 0;
        NeuralNetwork neuralNetwork(netDoc, config.parallelSequences(), maxSeqLength, inputSize, config.precision(), config.device() + (sameDevice ? 0 : dp.rank), &wi);
        if (config.deterministic() >= 0) hipCheck(cn_ctx_set_option(neuralNetwork.context(), "deterministic", config.deterministic()), neuralNetwork.context());
        if (dp.active) {
            // rendezvous: rank 0 draws the id and hands it to the other ranks through their pipes, then every rank joins
            char id[CN_COMM_ID_BYTES];
            if (dp.rank == 0) {
                hipCheck(cn_comm_unique_id(id), neuralNetwork.context());
                for (int fd : dp.idWriteFds)
                    if (write(fd, id, sizeof(id)) != (ssize_t)sizeof(id)) throw std::runtime_error("Could not hand the communicator id to a rank");
            } else {
                size_t got = 0;
                while (got < sizeof(id)) {
                    ssize_t n = read(dp.idReadFd, id + got, sizeof(id) - got);
                    if (n <= 0) throw std::runtime_error("Could not read the communicator id from rank 0");
                    got += (size_t)n;
                }
            }
            neuralNetwork.initDataParallel(id, dp.rank, dp.world);
            // test hook (tests/test_host_driver.py): this rank gives up after the rendezvous, the others are in their first exchange
            if (testHook("CN_DP_TEST_FAIL_RANK") && atoi(testHook("CN_DP_TEST_FAIL_RANK")) == dp.rank)
                throw std::runtime_error("test hook CN_DP_TEST_FAIL_RANK: this rank fails on purpose");
        }
        if (!trainingSet->empty() && trainingSet->outputPatternSize() != neuralNetwork.postOutputLayer().size())
            throw std::runtime_error("Post output layer size != target pattern size of the training set");
        if (!validationSet->empty() && validationSet->outputPatternSize() != neuralNetwork.postOutputLayer().size())
            throw std::runtime_error("Post output layer size != target pattern size of the validation set");
        if (!testSet->empty() && testSet->outputPatternSize() != neuralNetwork.postOutputLayer().size())
            throw std::runtime_error("Post output layer size != target pattern size of the test set");
        printf("done.\nLayers:\n");
        printLayers(neuralNetwork);
        printf("\n");

        const bool classificationTask = dynamic_cast<layers::MulticlassClassificationLayer *>(&neuralNetwork.postOutputLayer()) != 0 ||
                                        dynamic_cast<layers::BinaryClassificationLayer *>(&neuralNetwork.postOutputLayer()) != 0;   // main.cpp:165-166

        if (config.trainingMode()) {
            printf("Creating the optimizer... ");
            fflush(stdout);
            optimizers::SteepestDescentOptimizer optimizer(neuralNetwork, *trainingSet, *validationSet, *testSet, config.maxEpochs(),
                                                           config.maxEpochsNoBest(), config.validateEvery(), config.testEvery(),
                                                           config.learningRate(), config.momentum(), config.hybridOnlineBatch());
            printf("done.\n");
            printf("Optimizer type: Steepest descent with momentum\n");                         // main.cpp:660-678
            printf("Max training epochs:       %d\n", config.maxEpochs());
            printf("Max epochs until new best: %d\n", config.maxEpochsNoBest());
            printf("Validation error every:    %d\n", config.validateEvery());
            printf("Test error every:          %d\n", config.testEvery());
            printf("Learning rate:             %g\n", (double)config.learningRate());
            printf("Momentum:                  %g\n\n", (double)config.momentum());
            optimizer.setWeightNoise(config.weightNoiseSigma(), config.randomSeed());

            std::string infoRows;
            if (!config.continueFile().empty()) {                                               // main.cpp:198-204
                printf("Restoring state from '%s'... ", config.continueFile().c_str());
                fflush(stdout);
                restoreState(netDoc, &optimizer, &infoRows);
                printf("done.\n\n");
            }

            printf("Starting training...\n\n");
            printf(" Epoch | Duration |  Training error  | Validation error |    Test error    | New best \n");
            printf("-------+----------+------------------+------------------+------------------+----------\n");
            std::cout << infoRows;
            bool finished = false;
            while (!finished) {
                const char *errFormat = (classificationTask ? "%6.2lf%%%10.3lf |" : "%17.3lf |");
                const char *errSpace = "                  |";
                infoRows += printfRow(" %5d | ", optimizer.currentEpoch() + 1);
                auto t0 = std::chrono::steady_clock::now();
                finished = optimizer.train();
                double duration = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                infoRows += printfRow("%8.1lf |", duration);
                // CN_DRIVER_TIMING=1: the epoch's wall time with microsecond resolution on stderr (the table above keeps the
                // reference's format, one decimal); bench.py's driver leg reads it
                if (getenv("CN_DRIVER_TIMING")) fprintf(stderr, "TIMING epoch %d %.6f s %d frames\n", optimizer.currentEpoch(), duration, trainingSet->totalTimesteps());
                if (classificationTask) infoRows += printfRow(errFormat, (double)optimizer.curTrainingClassError() * 100.0, (double)optimizer.curTrainingError());
                else infoRows += printfRow(errFormat, (double)optimizer.curTrainingError());
                const bool validated = !validationSet->empty() && optimizer.currentEpoch() % config.validateEvery() == 0;
                if (validated) {
                    if (classificationTask) infoRows += printfRow(errFormat, (double)optimizer.curValidationClassError() * 100.0, (double)optimizer.curValidationError());
                    else infoRows += printfRow(errFormat, (double)optimizer.curValidationError());
                } else infoRows += printfRow("%s", errSpace);
                if (!testSet->empty() && optimizer.currentEpoch() % config.testEvery() == 0) {
                    if (classificationTask) infoRows += printfRow(errFormat, (double)optimizer.curTestClassError() * 100.0, (double)optimizer.curTestError());
                    else infoRows += printfRow(errFormat, (double)optimizer.curTestError());
                } else infoRows += printfRow("%s", errSpace);
                if (validated) {
                    const bool best = optimizer.epochsSinceLowestValidationError() == 0;
                    infoRows += printfRow(best ? "  yes   \n" : "  no    \n");
                    if (best && config.autosaveBest()) {                                         // main.cpp:254-268
                        std::string base = config.autosavePrefix();
                        if (base.empty()) {
                            size_t pos = config.networkFile().find_last_of('.');
                            base = (pos != std::string::npos && pos > 0) ? config.networkFile().substr(0, pos) : config.networkFile();
                        }
                        if (root) saveNetwork(neuralNetwork, base + ".best.jsn");
                    }
                } else infoRows += printfRow("        \n");
                if (config.autosave() && root) saveState(config, neuralNetwork, optimizer, infoRows);  // main.cpp:275-277
            }
            printf("\n");
            if (optimizer.epochsSinceLowestValidationError() == config.maxEpochsNoBest())
                printf("No new lowest error since %d epochs. Training stopped.\n", config.maxEpochsNoBest());
            else printf("Maximum number of training epochs reached. Training stopped.\n");
            if (!validationSet->empty()) printf("Lowest validation error: %lf\n", optimizer.lowestValidationError());
            else printf("Final training set error: %lf\n", optimizer.curTrainingError());
            printf("\n");
            printf("Storing the trained network in '%s'... ", config.trainedNetworkFile().c_str());
            if (root) saveNetwork(neuralNetwork, config.trainedNetworkFile());
            printf("done.\n");
        } else {
            feedForward(config, neuralNetwork, *feedForwardSet);
        }
    } catch (const std::exception &e) {
        // the console protocol of the reference (main.cpp:488-491) on stdout; a data-parallel rank also says so on stderr, which
        // every rank shares: ranks > 0 have no stdout, and the parent only sees exit codes
        printf("FAILED: %s\n", e.what());
        if (dp.active) { fprintf(stderr, "rank %d: FAILED: %s\n", dp.rank, e.what()); fflush(stderr); }
        return 2;
    }
    return 0;
}

// --gpus N: one process per GPU.  The ranks are forked HERE, before this process has made any GPU call (a process that
// has initialised the GPU must neither fork workers nor exec); each child runs trainerMain on device --device + rank with
// its shard of every fraction.  Rank 0 keeps the console; the other ranks' output is dropped.  The parent only waits:
// exit code = the first failing rank's (the others are terminated, they would wait in a collective for ever).
int runDataParallel(const Configuration &config, int world)
{
    if (testHook("CN_DP_SAME_DEVICE"))
        printf("Data-parallel training with %d ranks on device %d (CN_DP_SAME_DEVICE: a test mode), %d parallel sequences per rank.\n", world, config.device(), config.parallelSequences());
    else
    printf("Data-parallel training on %d GPU%s (devices %d..%d), %d parallel sequences per GPU.\n", world, world == 1 ? "" : "s",
           config.device(), config.device() + world - 1, config.parallelSequences());
    fflush(stdout);
    std::vector<int> readFd(world, -1), writeFd(world, -1);
    for (int r = 1; r < world; ++r) {
        int p[2];
        if (pipe(p) != 0) throw std::runtime_error("pipe() failed");
        readFd[r] = p[0]; writeFd[r] = p[1];
    }
    std::vector<pid_t> pids(world, -1);
    for (int r = 0; r < world; ++r) {
        pid_t pid = fork();
        if (pid < 0) {
            // the ranks already started would wait for the missing ones in the rendezvous for ever
            for (int k = 0; k < r; ++k) { kill(pids[k], SIGTERM); waitpid(pids[k], nullptr, 0); }
            throw std::runtime_error("fork() failed for rank " + std::to_string(r));
        }
        if (pid == 0) {
            prctl(PR_SET_PDEATHSIG, SIGTERM);            // a rank does not outlive the launcher (SIGKILL / SIGTERM of the parent)
            if (getppid() == 1) _exit(2);                // (the parent died between fork and prctl)
            DataParallel dp;
            dp.rank = r; dp.world = world; dp.active = true;
            if (r == 0) { for (int k = 1; k < world; ++k) { dp.idWriteFds.push_back(writeFd[k]); close(readFd[k]); } }
            else {
                dp.idReadFd = readFd[r];
                for (int k = 1; k < world; ++k) { close(writeFd[k]); if (k != r) close(readFd[k]); }
                if (!freopen("/dev/null", "w", stdout)) _exit(2);
            }
            int rc = trainerMain(config, dp);
            fflush(stdout);
            _exit(rc);
        }
        pids[r] = pid;
    }
    for (int r = 1; r < world; ++r) { close(readFd[r]); close(writeFd[r]); }
    int rc = 0, left = world;
    while (left > 0) {
        int status = 0;
        pid_t done = wait(&status);
        if (done < 0) break;
        --left;
        const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 2;
        int which = -1;
        for (int r = 0; r < world; ++r) if (pids[r] == done) { which = r; pids[r] = -1; }     // reaped: never signalled again
        if (code != 0) { fprintf(stderr, "rank %d exited with code %d\n", which, code); fflush(stderr); }
        if (code != 0 && rc == 0) {
            rc = code;
            for (int r = 0; r < world; ++r) if (pids[r] > 0) kill(pids[r], SIGTERM);
        }
    }
    return rc;
}

}  // namespace

int main(int argc, const char *argv[])
{
    try {
        Configuration config(argc, argv);
        if (config.help()) { fputs(Configuration::usage(), stdout); return 0; }
        if (config.listDevices()) {                                                              // main.cpp:509-525
            const int count = cn_device_count();
            std::cout << count << " devices found" << std::endl;
            for (int i = 0; i < count; ++i) {
                char name[256];
                if (cn_device_name(i, name, sizeof(name)) != CN_OK) { std::cerr << "FAILED: " << cn_last_error(0) << std::endl; return 2; }
                std::cout << i << ": " << name << std::endl;
            }
            return 0;
        }
        printf("Started in %s training mode.\n", config.hybridOnlineBatch() ? "hybrid online/batch" : "batch");   // Configuration.cpp:316
        printf("Computations run on the MI355X (libcurrennt_hip: %s, %s operands).\n", cn_version(),
               config.precision() == CN_PREC_BF16 ? "bf16" : (config.precision() == CN_PREC_BF16X3 ? "fp32 (split-bf16 x3 products)" : "fp32"));
        const bool forceDp = testHook("CN_DP_FORCE") != 0;       // test hook: --gpus 1 through the whole data-parallel path
        if (config.gpus() == 1 && !forceDp) return trainerMain(config);
        if (!config.trainingMode()) throw std::runtime_error("--gpus > 1 is a training option");
        return runDataParallel(config, config.gpus());
    } catch (const std::exception &e) {
        printf("FAILED: %s\n", e.what());
        return 2;
    }
}
