// Drives the functors of the reference's layers/LstmLayer.cu (see ref_common.h).  Signatures equal orc_lstm_forward /
// orc_lstm_backward of oracle/currennt_oracle.c so that tests swap one for the other.
#include "/root/reference/currennt_lib/src/layers/LstmLayer.cu"
#include "ref_common.h"

namespace {

typedef Cpu::real_vector rv;
typedef helpers::Matrix<Cpu> Mat;

enum { B_TMPOUT, B_TMPERR, B_CELL, B_CELLERR, B_NIACT, B_IGACT, B_FGACT, B_OGACT, B_NIDELTA, B_IGDELTA, B_FGDELTA, B_OGDELTA, B_COUNT };

// the per-direction vectors of LstmLayer.hpp:88-100 and the matrix views of LstmLayer.cu:577-628
struct Dir {
    rv b[B_COUNT];
    Mat niInput, igInput, fgInput, ogInput, niInternal, igInternal, fgInternal, ogInternal;
};
struct Layer {
    int P, L, H, dirs, PS, maxT, T, Tmin;
    real_t bias;
    rv weights, plOutputs;
    Cpu::pattype_vector pat;
    Dir d[2];
    const real_t *niBias, *igBias, *fgBias, *ogBias, *igPeep, *fgPeep, *ogPeep;
};

void bind(Layer &l, int P, int L, int bidir, real_t bias, int PS, int maxT, int T, int Tmin, const char *patTypes,
          const real_t *w, const real_t *x, const real_t *bufs)
{
    l.P = P; l.L = L; l.dirs = bidir ? 2 : 1; l.H = L / l.dirs; l.PS = PS; l.maxT = maxT; l.T = T; l.Tmin = Tmin; l.bias = bias;
    const int ls = L, pls = P, els = l.H;
    const int nw = ls * (4 * (pls + 1) + 4 * els + 3);                        // LstmLayer.cu:525, TrainableLayer.cu:101
    l.weights.assign(w, w + nw);
    l.plOutputs.assign(x, x + (size_t)T * PS * P);
    l.pat.assign(patTypes, patTypes + (size_t)T * PS);
    const size_t per = (size_t)PS * maxT * els;
    for (int k = 0; k < l.dirs; ++k)
        for (int i = 0; i < B_COUNT; ++i) l.d[k].b[i].assign(bufs + ((size_t)k * B_COUNT + i) * per, bufs + ((size_t)k * B_COUNT + i + 1) * per);
    real_t *raw = helpers::getRawPointer(l.weights);                          // :535-541
    l.niBias = raw + 4 * ls * pls + 0 * ls; l.igBias = raw + 4 * ls * pls + 1 * ls;
    l.fgBias = raw + 4 * ls * pls + 2 * ls; l.ogBias = raw + 4 * ls * pls + 3 * ls;
    l.igPeep = raw + 4 * ls * pls + 4 * ls + 4 * ls * ls / l.dirs + 0 * ls;
    l.fgPeep = raw + 4 * ls * pls + 4 * ls + 4 * ls * ls / l.dirs + 1 * ls;
    l.ogPeep = raw + 4 * ls * pls + 4 * ls + 4 * ls * ls / l.dirs + 2 * ls;
    for (int k = 0; k < l.dirs; ++k) {                                         // :583-596
        const int numInputWeights = ls * pls, numInternalWeights = ls * els;
        const int inputWeightsStart = (k == 1) ? numInputWeights / 2 : 0;
        const int internalWeightsStart = ((k == 1) ? numInternalWeights / 2 : 0) + 4 * (ls * (pls + 1));
        Dir &d = l.d[k];
        d.niInput = Mat(&l.weights, pls, els, inputWeightsStart + 0 * numInputWeights);
        d.igInput = Mat(&l.weights, pls, els, inputWeightsStart + 1 * numInputWeights);
        d.fgInput = Mat(&l.weights, pls, els, inputWeightsStart + 2 * numInputWeights);
        d.ogInput = Mat(&l.weights, pls, els, inputWeightsStart + 3 * numInputWeights);
        d.niInternal = Mat(&l.weights, els, els, internalWeightsStart + 0 * numInternalWeights);
        d.igInternal = Mat(&l.weights, els, els, internalWeightsStart + 1 * numInternalWeights);
        d.fgInternal = Mat(&l.weights, els, els, internalWeightsStart + 2 * numInternalWeights);
        d.ogInternal = Mat(&l.weights, els, els, internalWeightsStart + 3 * numInternalWeights);
    }
}
void unbind(const Layer &l, real_t *bufs)
{
    const size_t per = (size_t)l.PS * l.maxT * l.H;
    for (int k = 0; k < l.dirs; ++k)
        for (int i = 0; i < B_COUNT; ++i) memcpy(bufs + ((size_t)k * B_COUNT + i) * per, helpers::getRawPointer(l.d[k].b[i]), per * sizeof(real_t));
}
// timestep view of LstmLayer.cu:600-617
Mat ts(rv &v, const Layer &l, int timestep) { return Mat(&v, l.H, l.PS, timestep * l.H * l.PS); }

}  // namespace

// LstmLayer<Cpu>::computeForwardPass, LstmLayer.cu:763-886
REF_API void ref_lstm_forward(int P, int L, int bidir, real_t bias, int PS, int maxT, int T, int Tmin, const char *patTypes,
                              const real_t *w, const real_t *x, real_t *y, real_t *bufs)
{
    Layer l;
    bind(l, P, L, bidir, bias, PS, maxT, T, Tmin, patTypes, w, x, bufs);
    const int N = T * PS;
    Mat plOut(&l.plOutputs, P, N);                                             // :758
    for (int k = 0; k < l.dirs; ++k) {                                         // :771-786
        Dir &d = l.d[k];
        Mat(&d.b[B_NIACT], l.H, N).assignProduct(d.niInput, true, plOut, false);
        Mat(&d.b[B_IGACT], l.H, N).assignProduct(d.igInput, true, plOut, false);
        Mat(&d.b[B_FGACT], l.H, N).assignProduct(d.fgInput, true, plOut, false);
        Mat(&d.b[B_OGACT], l.H, N).assignProduct(d.ogInput, true, plOut, false);
    }
    const int els = l.H, n = PS * els;
    internal::ComputeBlockOutputFn fn;                                         // :794-810
    fn.effLayerSize = els; fn.prevOutputDistance = -n; fn.bias = bias; fn.patTypes = helpers::getRawPointer(l.pat);
    fn.niBiasWeights = l.niBias; fn.igBiasWeights = l.igBias; fn.fgBiasWeights = l.fgBias; fn.ogBiasWeights = l.ogBias;
    fn.igPeepWeights = l.igPeep; fn.fgPeepWeights = l.fgPeep; fn.ogPeepWeights = l.ogPeep;
    {
        Dir &d = l.d[0];
        fn.cellStates = helpers::getRawPointer(d.b[B_CELL]);
        fn.niActs = helpers::getRawPointer(d.b[B_NIACT]); fn.igActs = helpers::getRawPointer(d.b[B_IGACT]);
        fn.fgActs = helpers::getRawPointer(d.b[B_FGACT]); fn.ogActs = helpers::getRawPointer(d.b[B_OGACT]);
        for (int timestep = 0; timestep < T; ++timestep) {                     // :812-829
            if (timestep != 0) {
                Mat prev = ts(d.b[B_TMPOUT], l, timestep - 1);
                ts(d.b[B_NIACT], l, timestep).addProduct(d.niInternal, true, prev, false);
                ts(d.b[B_IGACT], l, timestep).addProduct(d.igInternal, true, prev, false);
                ts(d.b[B_FGACT], l, timestep).addProduct(d.fgInternal, true, prev, false);
                ts(d.b[B_OGACT], l, timestep).addProduct(d.ogInternal, true, prev, false);
            }
            thrust::transform(thrust::counting_iterator<int>(n * timestep), thrust::counting_iterator<int>(n * timestep) + n,
                              thrust::make_zip_iterator(thrust::make_tuple(thrust::constant_iterator<bool>(!timestep),
                                                                           thrust::constant_iterator<bool>(timestep >= Tmin))),
                              d.b[B_TMPOUT].begin() + n * timestep, fn);
        }
    }
    if (bidir) {                                                               // :832-865
        Dir &d = l.d[1];
        fn.prevOutputDistance = +n;
        fn.niBiasWeights += els; fn.igBiasWeights += els; fn.fgBiasWeights += els; fn.ogBiasWeights += els;
        fn.igPeepWeights += els; fn.fgPeepWeights += els; fn.ogPeepWeights += els;
        fn.cellStates = helpers::getRawPointer(d.b[B_CELL]);
        fn.niActs = helpers::getRawPointer(d.b[B_NIACT]); fn.igActs = helpers::getRawPointer(d.b[B_IGACT]);
        fn.fgActs = helpers::getRawPointer(d.b[B_FGACT]); fn.ogActs = helpers::getRawPointer(d.b[B_OGACT]);
        for (int timestep = T - 1; timestep >= 0; --timestep) {
            if (timestep != T - 1) {
                Mat prev = ts(d.b[B_TMPOUT], l, timestep + 1);
                ts(d.b[B_NIACT], l, timestep).addProduct(d.niInternal, true, prev, false);
                ts(d.b[B_IGACT], l, timestep).addProduct(d.igInternal, true, prev, false);
                ts(d.b[B_FGACT], l, timestep).addProduct(d.fgInternal, true, prev, false);
                ts(d.b[B_OGACT], l, timestep).addProduct(d.ogInternal, true, prev, false);
            }
            thrust::transform(thrust::counting_iterator<int>(n * timestep), thrust::counting_iterator<int>(n * timestep) + n,
                              thrust::make_zip_iterator(thrust::make_tuple(thrust::constant_iterator<bool>(timestep == T - 1),
                                                                           thrust::constant_iterator<bool>(timestep >= Tmin))),
                              d.b[B_TMPOUT].begin() + n * timestep, fn);
        }
    }
    rv out((size_t)N * L);
    if (bidir) {                                                               // :869-882
        internal::ResortOutputsFn rf;
        rf.layerSize = L; rf.effLayerSize = L / 2;
        rf.fwOutputs = helpers::getRawPointer(l.d[0].b[B_TMPOUT]); rf.bwOutputs = helpers::getRawPointer(l.d[1].b[B_TMPOUT]);
        thrust::transform(thrust::counting_iterator<int>(0), thrust::counting_iterator<int>(0) + N * L, out.begin(), rf);
    } else {
        thrust::copy(l.d[0].b[B_TMPOUT].begin(), l.d[0].b[B_TMPOUT].begin() + (size_t)N * L, out.begin());   // :883-885 (a swap there)
    }
    memcpy(y, helpers::getRawPointer(out), (size_t)N * L * sizeof(real_t));
    unbind(l, bufs);
}

// LstmLayer<Cpu>::computeBackwardPass, LstmLayer.cu:888-1051
REF_API void ref_lstm_backward(int P, int L, int bidir, real_t bias, int PS, int maxT, int T, int Tmin, const char *patTypes,
                               const real_t *w, const real_t *x, const real_t *outErr, real_t *prevErr, real_t *wu, real_t *bufs)
{
    Layer l;
    bind(l, P, L, bidir, bias, PS, maxT, T, Tmin, patTypes, w, x, bufs);
    const int N = T * PS, els = l.H, n = PS * els;
    rv outputErrors(outErr, outErr + (size_t)N * L);
    if (bidir) {                                                               // :892-905
        internal::ResortOutputErrorsFn rf;
        rf.layerSize = L; rf.effLayerSize = L / 2;
        rf.fwOutputErrors = helpers::getRawPointer(l.d[0].b[B_TMPERR]); rf.bwOutputErrors = helpers::getRawPointer(l.d[1].b[B_TMPERR]);
        const int cnt = N * L;
        thrust::for_each(thrust::make_zip_iterator(thrust::make_tuple(outputErrors.begin(), thrust::counting_iterator<int>(0))),
                         thrust::make_zip_iterator(thrust::make_tuple(outputErrors.begin() + cnt, thrust::counting_iterator<int>(0) + cnt)), rf);
    } else {
        thrust::copy(outputErrors.begin(), outputErrors.end(), l.d[0].b[B_TMPERR].begin());      // :907-908 (swaps there)
    }
    internal::ComputeBlockErrorsFn fn;                                         // :916-933
    fn.effLayerSize = els; fn.prevOutputDistance = -n; fn.patTypes = helpers::getRawPointer(l.pat);
    fn.igPeepWeights = l.igPeep; fn.fgPeepWeights = l.fgPeep; fn.ogPeepWeights = l.ogPeep;
    auto point = [&](Dir &d) {
        fn.cellStates = helpers::getRawPointer(d.b[B_CELL]);
        fn.niActs = helpers::getRawPointer(d.b[B_NIACT]); fn.igActs = helpers::getRawPointer(d.b[B_IGACT]);
        fn.fgActs = helpers::getRawPointer(d.b[B_FGACT]); fn.ogActs = helpers::getRawPointer(d.b[B_OGACT]);
        fn.cellStateErrors = helpers::getRawPointer(d.b[B_CELLERR]);
        fn.niDeltas = helpers::getRawPointer(d.b[B_NIDELTA]); fn.igDeltas = helpers::getRawPointer(d.b[B_IGDELTA]);
        fn.fgDeltas = helpers::getRawPointer(d.b[B_FGDELTA]); fn.ogDeltas = helpers::getRawPointer(d.b[B_OGDELTA]);
    };
    {
        Dir &d = l.d[0];
        point(d);
        for (int timestep = T - 1; timestep >= 0; --timestep) {                // :935-951
            if (timestep != T - 1) {
                Mat e = ts(d.b[B_TMPERR], l, timestep);
                e.addProduct(d.niInternal, false, ts(d.b[B_NIDELTA], l, timestep + 1), false);
                e.addProduct(d.igInternal, false, ts(d.b[B_IGDELTA], l, timestep + 1), false);
                e.addProduct(d.fgInternal, false, ts(d.b[B_FGDELTA], l, timestep + 1), false);
                e.addProduct(d.ogInternal, false, ts(d.b[B_OGDELTA], l, timestep + 1), false);
            }
            thrust::for_each(
                thrust::make_zip_iterator(thrust::make_tuple(d.b[B_TMPERR].begin() + n * timestep, thrust::counting_iterator<int>(n * timestep),
                                                             thrust::constant_iterator<bool>(timestep == T - 1), thrust::constant_iterator<bool>(!timestep),
                                                             thrust::constant_iterator<bool>(timestep >= Tmin))),
                thrust::make_zip_iterator(thrust::make_tuple(d.b[B_TMPERR].begin() + n * timestep + n, thrust::counting_iterator<int>(n * timestep) + n,
                                                             thrust::constant_iterator<bool>(timestep == T - 1) + n, thrust::constant_iterator<bool>(!timestep) + n,
                                                             thrust::constant_iterator<bool>(timestep >= Tmin) + n)),
                fn);
        }
    }
    if (bidir) {                                                               // :954-986
        Dir &d = l.d[1];
        fn.prevOutputDistance = +n;
        fn.igPeepWeights += els; fn.fgPeepWeights += els; fn.ogPeepWeights += els;
        point(d);
        for (int timestep = 0; timestep < T; ++timestep) {
            if (timestep != 0) {
                Mat e = ts(d.b[B_TMPERR], l, timestep);
                e.addProduct(d.niInternal, false, ts(d.b[B_NIDELTA], l, timestep - 1), false);
                e.addProduct(d.igInternal, false, ts(d.b[B_IGDELTA], l, timestep - 1), false);
                e.addProduct(d.fgInternal, false, ts(d.b[B_FGDELTA], l, timestep - 1), false);
                e.addProduct(d.ogInternal, false, ts(d.b[B_OGDELTA], l, timestep - 1), false);
            }
            thrust::for_each(
                thrust::make_zip_iterator(thrust::make_tuple(d.b[B_TMPERR].begin() + n * timestep, thrust::counting_iterator<int>(n * timestep),
                                                             thrust::constant_iterator<bool>(!timestep), thrust::constant_iterator<bool>(timestep == T - 1),
                                                             thrust::constant_iterator<bool>(timestep >= Tmin))),
                thrust::make_zip_iterator(thrust::make_tuple(d.b[B_TMPERR].begin() + n * timestep + n, thrust::counting_iterator<int>(n * timestep) + n,
                                                             thrust::constant_iterator<bool>(!timestep) + n, thrust::constant_iterator<bool>(timestep == T - 1) + n,
                                                             thrust::constant_iterator<bool>(timestep >= Tmin) + n)),
                fn);
        }
    }
    if (prevErr) {                                                             // :990-1009
        rv plErrors((size_t)N * P);
        Mat plErrorsMatrix(&plErrors, P, N);
        Dir &f = l.d[0];
        plErrorsMatrix.assignProduct(f.niInput, false, Mat(&f.b[B_NIDELTA], els, N), false);
        plErrorsMatrix.addProduct(f.igInput, false, Mat(&f.b[B_IGDELTA], els, N), false);
        plErrorsMatrix.addProduct(f.fgInput, false, Mat(&f.b[B_FGDELTA], els, N), false);
        plErrorsMatrix.addProduct(f.ogInput, false, Mat(&f.b[B_OGDELTA], els, N), false);
        if (bidir) {
            Dir &b = l.d[1];
            plErrorsMatrix.addProduct(b.niInput, false, Mat(&b.b[B_NIDELTA], els, N), false);
            plErrorsMatrix.addProduct(b.igInput, false, Mat(&b.b[B_IGDELTA], els, N), false);
            plErrorsMatrix.addProduct(b.fgInput, false, Mat(&b.b[B_FGDELTA], els, N), false);
            plErrorsMatrix.addProduct(b.ogInput, false, Mat(&b.b[B_OGDELTA], els, N), false);
        }
        memcpy(prevErr, helpers::getRawPointer(plErrors), (size_t)N * P * sizeof(real_t));
    }
    {                                                                          // :1012-1044
        internal::ComputeWeightUpdateFn wf;
        wf.layerSize = L; wf.effLayerSize = els; wf.precLayerSize = P;
        wf.timestepDistance = PS * L / l.dirs; wf.parallelSequences = PS; wf.patternsCount = T * PS;
        wf.biasWeightsOffset = L * P * 4; wf.internalWeightsOffset = wf.biasWeightsOffset + L * 4;
        wf.peepholeWeightsOffset = wf.internalWeightsOffset + L * els * 4;
        wf.bias = bias;
        wf.plOutputs = helpers::getRawPointer(l.plOutputs);
        Dir &f = l.d[0], &b = l.d[1];                                          // (uni: the bw vectors are empty, as in the reference)
        wf.fwOutputs = helpers::getRawPointer(f.b[B_TMPOUT]); wf.bwOutputs = helpers::getRawPointer(b.b[B_TMPOUT]);
        wf.fwCellStates = helpers::getRawPointer(f.b[B_CELL]); wf.bwCellStates = helpers::getRawPointer(b.b[B_CELL]);
        wf.fwNiDeltas = helpers::getRawPointer(f.b[B_NIDELTA]); wf.bwNiDeltas = helpers::getRawPointer(b.b[B_NIDELTA]);
        wf.fwIgDeltas = helpers::getRawPointer(f.b[B_IGDELTA]); wf.bwIgDeltas = helpers::getRawPointer(b.b[B_IGDELTA]);
        wf.fwFgDeltas = helpers::getRawPointer(f.b[B_FGDELTA]); wf.bwFgDeltas = helpers::getRawPointer(b.b[B_FGDELTA]);
        wf.fwOgDeltas = helpers::getRawPointer(f.b[B_OGDELTA]); wf.bwOgDeltas = helpers::getRawPointer(b.b[B_OGDELTA]);
        const int nw = (int)l.weights.size();
        rv weightUpdates(nw);
        thrust::transform(thrust::counting_iterator<int>(0), thrust::counting_iterator<int>(0) + nw, weightUpdates.begin(), wf);
        memcpy(wu, helpers::getRawPointer(weightUpdates), (size_t)nw * sizeof(real_t));
    }
    unbind(l, bufs);
}
