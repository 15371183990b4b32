#include "DataSet.hpp"

#include <algorithm>
#include <limits>
#include <stdexcept>

#include "../NetCdf3.hpp"

namespace currennt_hip {
namespace data_sets {

DataSet::DataSet() {}

// --truncate_seq (DataSet.cpp:527-542): a sequence is cut into pieces of `trunc` steps for as long as MORE than 1.5 x trunc
// steps remain; what remains then (0.5 .. 1.5 x trunc steps) is the last piece.  trunc <= 0: one piece.
std::vector<int> DataSet::truncatedPieces(int length, int trunc)
{
    std::vector<int> pieces;
    if (trunc > 0)
        for (; length > 1.5 * trunc; length -= trunc) pieces.push_back(trunc);
    if (length > 0) pieces.push_back(length);
    return pieces;
}

DataSet::~DataSet()
{
    if (m_prefetch.valid()) m_prefetch.wait();
}

DataSet::DataSet(const std::vector<std::string> &ncfiles, int parSeq, real_t fraction, int truncSeqLength,
                 bool fracShuf, bool seqShuf, bool sortByLength, unsigned randomSeed, const Augment &augment)
    : m_fractionShuffling(fracShuf), m_sequenceShuffling(seqShuf), m_parallelSequences(parSeq),
      m_minSeqLength(std::numeric_limits<int>::max()), m_maxSeqLength(std::numeric_limits<int>::min()),
      m_rngState(randomSeed ? randomSeed : 5489u), m_augment(augment), m_noiseGen(randomSeed)
{
    if (augment.contextLeft < 0 || augment.contextRight < 0 || augment.outputLag < 0)
        throw std::runtime_error("Negative input context / output time lag");
    if (fraction <= 0 || fraction > 1) throw std::runtime_error("Invalid fraction");        // DataSet.cpp:457-458
    bool firstFile = true;
    for (size_t fi = 0; fi < ncfiles.size(); ++fi) {
        NetCdf3File nc(ncfiles[fi]);
        const int maxSeqTagLength = nc.dimension("maxSeqTagLength");
        if (firstFile) {                                                                       // :489-500
            m_isClassificationData = nc.hasDimension("numLabels");
            m_inputPatternSize = nc.dimension("inputPattSize");
            if (m_isClassificationData) { int numLabels = nc.dimension("numLabels"); m_outputPatternSize = (numLabels == 2 ? 1 : numLabels); }
            else m_outputPatternSize = nc.dimension("targetPattSize");
        } else {                                                                               // :502-515
            if (m_isClassificationData) {
                if (!nc.hasDimension("numLabels")) throw std::runtime_error("Cannot combine classification with regression NC");
                int numLabels = nc.dimension("numLabels");
                if (m_outputPatternSize != (numLabels == 2 ? 1 : numLabels)) throw std::runtime_error("Number of classes mismatch in NC files");
            } else if (m_outputPatternSize != nc.dimension("targetPattSize")) throw std::runtime_error("Number of targets mismatch in NC files");
            if (m_inputPatternSize != nc.dimension("inputPattSize")) throw std::runtime_error("Number of inputs mismatch in NC files");
        }
        int nSeq = nc.dimension("numSeqs");
        nSeq = std::max((int)((real_t)nSeq * fraction), 1);                                    // :518-520
        std::vector<int> lengths = nc.readInts("seqLengths", 0, nSeq);
        std::vector<sequence_t> sequences;
        for (int i = 0; i < nSeq; ++i) {
            m_totalTimesteps += lengths[i];
            std::string seqTag = nc.readString("seqTags", i, maxSeqTagLength);
            const std::vector<int> pieces = truncatedPieces(lengths[i], truncSeqLength);
            for (size_t k = 0; k < pieces.size(); ++k) {
                sequence_t seq;
                seq.originalSeqIdx = (int)k; seq.length = pieces[k]; seq.seqTag = seqTag;
                seq.inputsBegin = 0; seq.targetsBegin = 0;
                sequences.push_back(seq);
            }
        }
        size_t frames = 0;
        for (size_t i = 0; i < sequences.size(); ++i) frames += sequences[i].length;
        // bulk read (the reference reads sequence by sequence into its cache file, :549-575)
        size_t inBase = m_inputData.size();
        { std::vector<float> x = nc.readFloats("inputs", 0, frames * m_inputPatternSize); m_inputData.insert(m_inputData.end(), x.begin(), x.end()); }
        size_t tgBase;
        if (m_isClassificationData) {
            tgBase = m_classData.size();
            std::vector<int> t = nc.readInts("targetClasses", 0, frames); m_classData.insert(m_classData.end(), t.begin(), t.end());
        } else {
            tgBase = m_targetData.size();
            std::vector<float> t = nc.readFloats("targetPatterns", 0, frames * m_outputPatternSize); m_targetData.insert(m_targetData.end(), t.begin(), t.end());
        }
        size_t pos = 0;
        for (size_t i = 0; i < sequences.size(); ++i) {
            m_minSeqLength = std::min(m_minSeqLength, sequences[i].length);
            m_maxSeqLength = std::max(m_maxSeqLength, sequences[i].length);
            sequences[i].inputsBegin = inBase + pos * m_inputPatternSize;
            sequences[i].targetsBegin = tgBase + pos * (m_isClassificationData ? 1 : m_outputPatternSize);
            pos += sequences[i].length;
        }
        if (firstFile) {                                                                       // :577-588
            if (nc.hasVariable("outputMeans") && nc.hasVariable("outputStdevs")) {
                m_outputMeans = nc.readFloats("outputMeans", 0, m_outputPatternSize);
                m_outputStdevs = nc.readFloats("outputStdevs", 0, m_outputPatternSize);
            } else { m_outputMeans.assign(m_outputPatternSize, 0.0f); m_outputStdevs.assign(m_outputPatternSize, 1.0f); }
        }
        m_sequences.insert(m_sequences.end(), sequences.begin(), sequences.end());
        firstFile = false;
    }
    m_totalSequences = (int)m_sequences.size();
    if (sortByLength)                                                                          // :603-605 (std::sort, comp_seqs :164-167)
        std::sort(m_sequences.begin(), m_sequences.end(), [](const sequence_t &a, const sequence_t &b) { return a.length < b.length; });
}

void DataSet::setShard(int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("Invalid data-parallel shard");
    if (m_prefetch.valid()) m_prefetch.wait();
    m_rank = rank; m_world = world;
}

unsigned DataSet::nextRandom(unsigned n)
{
    // The reference draws from boost::mt19937 (DataSet.cpp:169-181), which is not available here; a
    // xorshift generator seeded from --random_seed keeps shuffles reproducible per seed.
    m_rngState ^= m_rngState << 13; m_rngState ^= m_rngState >> 17; m_rngState ^= m_rngState << 5;
    return n ? m_rngState % n : 0;
}

void DataSet::shuffleSequences()
{
    for (size_t i = m_sequences.size(); i > 1; --i) std::swap(m_sequences[i - 1], m_sequences[nextRandom((unsigned)i)]);
}

void DataSet::shuffleFractions()
{
    std::vector<std::vector<sequence_t> > fractions;
    for (size_t i = 0; i < m_sequences.size(); ++i) {
        if (i % ((size_t)m_parallelSequences * m_world) == 0) fractions.resize(fractions.size() + 1);      // whole global fractions move
        fractions.back().push_back(m_sequences[i]);
    }
    for (size_t i = fractions.size(); i > 1; --i) std::swap(fractions[i - 1], fractions[nextRandom((unsigned)i)]);
    m_sequences.clear();
    for (size_t i = 0; i < fractions.size(); ++i) m_sequences.insert(m_sequences.end(), fractions[i].begin(), fractions[i].end());
}

void DataSet::makeFraction(int firstSeqIdx, DataSetFraction *frac)
{
    const int PS = m_parallelSequences;
    const int ctxL = m_augment.contextLeft, ctxR = m_augment.contextRight, lag = m_augment.outputLag;   // :302-305
    const int P = m_inputPatternSize, Pf = P * (ctxL + ctxR + 1);
    *frac = DataSetFraction();
    frac->m_inputPatternSize = Pf;
    frac->m_outputPatternSize = m_outputPatternSize;
    frac->m_maxSeqLength = std::numeric_limits<int>::min();
    frac->m_minSeqLength = std::numeric_limits<int>::max();
    // this rank's sequences of the global fraction: firstSeqIdx + rank, + world, ... (world = 1: the reference's loop)
    auto globalIdx = [&](int i) { return firstSeqIdx + m_rank + i * m_world; };
    for (int i = 0; i < PS; ++i) {                                                             // :316-328
        const int seqIdx = globalIdx(i);
        if (seqIdx >= (int)m_sequences.size()) continue;
        frac->m_maxSeqLength = std::max(frac->m_maxSeqLength, m_sequences[seqIdx].length);
        frac->m_minSeqLength = std::min(frac->m_minSeqLength, m_sequences[seqIdx].length);
        DataSetFraction::seq_info_t si = { m_sequences[seqIdx].originalSeqIdx, m_sequences[seqIdx].length, m_sequences[seqIdx].seqTag };
        frac->m_seqInfo.push_back(si);
    }
    if (frac->m_seqInfo.empty()) { frac->m_maxSeqLength = 1; frac->m_minSeqLength = 0; }        // all-dummy fraction (setShard)
    const size_t slots = (size_t)frac->m_maxSeqLength * PS;
    frac->m_inputs.assign(slots * Pf, 0.0f);                                                   // :330-336
    frac->m_patTypes.assign(slots, (char)PATTYPE_NONE);
    if (m_isClassificationData) frac->m_targetClasses.assign(slots, -1);
    else frac->m_outputs.assign(slots * m_outputPatternSize, 0.0f);
    std::vector<float> noisy;
    for (int i = 0; i < PS; ++i) {
        if (globalIdx(i) >= (int)m_sequences.size()) continue;
        const sequence_t &seq = m_sequences[globalIdx(i)];
        const float *src = m_inputData.data() + seq.inputsBegin;
        if (m_augment.noiseDeviation) {                                                        // _addNoise, :250-265
            // the reference draws from boost::mt19937 + boost::normal_distribution (absent here);
            // std::mt19937 seeded from --random_seed keeps runs reproducible per seed
            std::normal_distribution<real_t> dist((real_t)0, m_augment.noiseDeviation);
            noisy.assign(src, src + (size_t)seq.length * P);
            for (size_t k = 0; k < noisy.size(); ++k) noisy[k] += dist(m_noiseGen);
            src = noisy.data();
        }
        for (int t = 0; t < seq.length; ++t) {
            const size_t slot = (size_t)t * PS + i;
            for (int off = -ctxL, k = 0; off <= ctxR; ++off, ++k) {                            // :346-366, edge frames are repeated
                const int ts = std::min(std::max(t + off, 0), seq.length - 1);
                std::copy(src + (size_t)ts * P, src + (size_t)(ts + 1) * P, frac->m_inputs.begin() + slot * Pf + (size_t)k * P);
            }
            if (m_isClassificationData)                                                        // :372-380, class 0 before the lag
                frac->m_targetClasses[slot] = t >= lag ? m_classData[seq.targetsBegin + (t - lag)] : 0;
            else if (t >= lag)                                                                 // :383-397, 1.0 before the lag
                std::copy(m_targetData.begin() + seq.targetsBegin + (size_t)(t - lag) * m_outputPatternSize,
                          m_targetData.begin() + seq.targetsBegin + (size_t)(t - lag + 1) * m_outputPatternSize,
                          frac->m_outputs.begin() + slot * m_outputPatternSize);
            else std::fill_n(frac->m_outputs.begin() + slot * m_outputPatternSize, m_outputPatternSize, 1.0f);
            frac->m_patTypes[slot] = (char)(t == 0 ? PATTYPE_FIRST : (t == seq.length - 1 ? PATTYPE_LAST : PATTYPE_NORMAL));   // :400-407
        }
    }
}

bool DataSet::produceNext(DataSetFraction *frac)
{
    if (m_curFirstSeqIdx == -1 || m_curFirstSeqIdx == 0) {       // start of an epoch: _makeFirstFractionTask, :416-427
        if (m_curFirstSeqIdx == -1) m_curFirstSeqIdx = 0;
        if (m_sequenceShuffling) shuffleSequences();
        if (m_fractionShuffling) shuffleFractions();
    }
    if (m_curFirstSeqIdx < (int)m_sequences.size()) {
        makeFraction(m_curFirstSeqIdx, frac);
        m_curFirstSeqIdx += m_parallelSequences * m_world;
        return true;
    }
    m_curFirstSeqIdx = 0;                                        // :660-662
    return false;
}

bool DataSet::getNextFraction(DataSetFraction *frac)
{
    // :632-668: take the fraction the worker prepared and start the worker on the one after it.  After the
    // last fraction the worker's task only rewinds (returns false); the next epoch's shuffles run when that
    // epoch's first fraction is asked for, as in the reference.
    if (!m_prefetch.valid()) m_prefetch = std::async(std::launch::async, [this] { return produceNext(&m_next); });
    const bool ok = m_prefetch.get();
    if (ok) {
        std::swap(*frac, m_next);
        m_prefetch = std::async(std::launch::async, [this] { return produceNext(&m_next); });
    }
    return ok;
}

}  // namespace data_sets
}  // namespace currennt_hip
