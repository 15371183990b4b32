// In-memory data set and fraction packer: the host objects of currennt_lib/src/data_sets/
// {DataSet,DataSetFraction}.{hpp,cpp} for NetCDF-3 classic files (the reference goes through
// libnetcdf and a /tmp cache file; sequences here are kept in RAM).
#pragma once

#include <future>
#include <random>
#include <string>
#include <vector>

#include "../Types.hpp"

namespace currennt_hip {
namespace data_sets {

// one mini batch of parallel sequences (DataSetFraction.hpp:40-66)
class DataSetFraction {
public:
    struct seq_info_t { int originalSeqIdx; int length; std::string seqTag; };

    int inputPatternSize() const { return m_inputPatternSize; }
    int outputPatternSize() const { return m_outputPatternSize; }
    int maxSeqLength() const { return m_maxSeqLength; }
    int minSeqLength() const { return m_minSeqLength; }
    int numSequences() const { return (int)m_seqInfo.size(); }
    const seq_info_t &seqInfo(int seqIdx) const { return m_seqInfo.at(seqIdx); }
    const Hip::pattype_vector &patTypes() const { return m_patTypes; }
    const Hip::real_vector &inputs() const { return m_inputs; }
    const Hip::real_vector &outputs() const { return m_outputs; }
    const Hip::int_vector &targetClasses() const { return m_targetClasses; }

private:
    friend class DataSet;
    int m_inputPatternSize = 0, m_outputPatternSize = 0, m_maxSeqLength = 0, m_minSeqLength = 0;
    std::vector<seq_info_t> m_seqInfo;
    Hip::real_vector m_inputs, m_outputs;
    Hip::pattype_vector m_patTypes;
    Hip::int_vector m_targetClasses;
};

// what the reference pulls from Configuration::instance() inside _makeFractionTask / _addNoise
// (DataSet.cpp:252-265,302-305): Gaussian input noise, context splicing, output time lag
struct Augment { real_t noiseDeviation = 0; int contextLeft = 0, contextRight = 0, outputLag = 0; };

class DataSet {
public:
    typedef data_sets::Augment Augment;
    struct sequence_t { int originalSeqIdx; int length; std::string seqTag; size_t inputsBegin; size_t targetsBegin; };

    DataSet();                                              // empty set (DataSet.cpp:429-442)
    // DataSet.cpp:443-606; `sortByLength` = Configuration::trainingMode() there (:603-605)
    DataSet(const std::vector<std::string> &ncfiles, int parSeq, real_t fraction, int truncSeqLength,
            bool fracShuf, bool seqShuf, bool sortByLength, unsigned randomSeed, const Augment &augment = Augment());
    ~DataSet();
    DataSet(const DataSet &) = delete;
    DataSet &operator=(const DataSet &) = delete;

    // Data-parallel training: a GLOBAL fraction is world * parallelSequences consecutive sequences of the
    // (sorted / shuffled) list and rank r packs sequences r, r + world, r + 2 world, ... of it -- round-robin over the
    // length-sorted list keeps the ranks' T nearly equal.  Every rank sees the same number of fractions per epoch; a rank
    // whose share of the last global fraction is empty gets an all-dummy fraction (one time step of PATTYPE_NONE), which
    // contributes zero error and zero gradients but keeps the collectives matched.  Shuffles use the same seed on every
    // rank, so all ranks permute alike.
    void setShard(int rank, int world);
    bool isClassificationData() const { return m_isClassificationData; }
    bool empty() const { return m_totalTimesteps == 0; }
    // next fraction, or false once per epoch after the last one (DataSet.cpp:632-668)
    bool getNextFraction(DataSetFraction *frac);
    int totalSequences() const { return m_totalSequences; }
    int totalTimesteps() const { return m_totalTimesteps; }
    int minSeqLength() const { return m_minSeqLength; }
    int maxSeqLength() const { return m_maxSeqLength; }
    int inputPatternSize() const { return m_inputPatternSize; }                        // per frame, before context splicing
    int fractionInputPatternSize() const { return m_inputPatternSize * (m_augment.contextLeft + m_augment.contextRight + 1); }
    int outputPatternSize() const { return m_outputPatternSize; }
    const Hip::real_vector &outputMeans() const { return m_outputMeans; }
    const Hip::real_vector &outputStdevs() const { return m_outputStdevs; }

private:
    bool m_fractionShuffling = false, m_sequenceShuffling = false, m_isClassificationData = false;
    int m_parallelSequences = 0, m_totalSequences = 0, m_totalTimesteps = 0, m_minSeqLength = 0, m_maxSeqLength = 0,
        m_inputPatternSize = 0, m_outputPatternSize = 0, m_curFirstSeqIdx = -1, m_rank = 0, m_world = 1;
    unsigned m_rngState = 0;
    Hip::real_vector m_outputMeans, m_outputStdevs, m_inputData, m_targetData;
    Hip::int_vector m_classData;
    std::vector<sequence_t> m_sequences;

    Augment m_augment;
    std::mt19937 m_noiseGen;
    // one fraction is packed ahead on a worker thread while the GPU runs the current one
    // (the reference's boost::thread hand-over, DataSet.cpp:589,632-668)
    std::future<bool> m_prefetch;
    DataSetFraction m_next;
    bool produceNext(DataSetFraction *frac);

    static std::vector<int> truncatedPieces(int length, int trunc);   // --truncate_seq, DataSet.cpp:527-542
    void makeFraction(int firstSeqIdx, DataSetFraction *frac);        // _makeFractionTask, DataSet.cpp:300-414
    void shuffleSequences();                                          // :226-230
    void shuffleFractions();                                          // :232-250
    unsigned nextRandom(unsigned n);
};

}  // namespace data_sets
}  // namespace currennt_hip
