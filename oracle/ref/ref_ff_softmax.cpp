// Drives the functors of the reference's layers/FeedForwardLayer.cu and layers/SoftmaxLayer.cu (see ref_common.h).
// Signatures equal orc_ff_* / orc_softmax_* of oracle/currennt_oracle.c.
#include "/root/reference/currennt_lib/src/layers/FeedForwardLayer.cu"
#include "/root/reference/currennt_lib/src/layers/SoftmaxLayer.cu"
#include "ref_common.h"

namespace {

typedef Cpu::real_vector rv;
typedef helpers::Matrix<Cpu> Mat;

// FeedForwardLayer<Cpu,TActFn>::computeForwardPass, FeedForwardLayer.cu:143-170
template <typename TActFn>
void ff_forward(int P, int L, real_t bias, int N, rv &weights, rv &plOutputs, rv &outputs)
{
    {
        Mat weightsMatrix(&weights, P, L), plOutputsMatrix(&plOutputs, P, N), outputsMatrix(&outputs, L, N);
        outputsMatrix.assignProduct(weightsMatrix, true, plOutputsMatrix, false);
    }
    internal::ComputeOutputFn<TActFn> fn;
    fn.layerSize = L; fn.bias = bias; fn.biasWeights = helpers::getRawPointer(weights) + L * P;
    thrust::transform(outputs.begin(), outputs.begin() + N * L, thrust::counting_iterator<int>(0), outputs.begin(), fn);
}

// FeedForwardLayer<Cpu,TActFn>::computeBackwardPass, FeedForwardLayer.cu:172-224
template <typename TActFn>
void ff_backward(int P, int L, real_t bias, int N, rv &weights, rv &plOutputs, rv &outputs, rv &outputErrors, rv *plErrors, rv &weightUpdates)
{
    {
        internal::ComputeDeltaFn<TActFn> fn;
        const int n = N * L;
        thrust::for_each(thrust::make_zip_iterator(thrust::make_tuple(outputErrors.begin(), outputs.begin())),
                         thrust::make_zip_iterator(thrust::make_tuple(outputErrors.begin() + n, outputs.begin() + n)), fn);
    }
    if (plErrors) {
        Mat weightsMatrix(&weights, P, L), plErrorsMatrix(plErrors, P, N), deltasMatrix(&outputErrors, L, N);
        plErrorsMatrix.assignProduct(weightsMatrix, false, deltasMatrix, false);
    }
    {
        Mat weightUpdatesMatrix(&weightUpdates, P, L), plOutputsMatrix(&plOutputs, P, N), deltasMatrix(&outputErrors, L, N);
        weightUpdatesMatrix.assignProduct(plOutputsMatrix, false, deltasMatrix, true);
    }
    {
        internal::ComputeBiasWeightUpdateFn fn;
        fn.layerSize = L; fn.patternsCount = N; fn.bias = bias; fn.deltas = helpers::getRawPointer(outputErrors);
        thrust::transform(thrust::counting_iterator<int>(0), thrust::counting_iterator<int>(0) + L, weightUpdates.begin() + P * L, fn);
    }
}

template <typename F> void by_act(int act, F &&f)
{
    if (act == 0) f(activation_functions::Tanh());
    else if (act == 1) f(activation_functions::Logistic());
    else f(activation_functions::Identity());
}

}  // namespace

REF_API void ref_ff_forward(int act, int P, int L, real_t bias, int N, const real_t *w, const real_t *x, real_t *y)
{
    rv weights(w, w + (size_t)L * (P + 1)), plOutputs(x, x + (size_t)N * P), outputs((size_t)N * L);
    by_act(act, [&](auto a) { ff_forward<decltype(a)>(P, L, bias, N, weights, plOutputs, outputs); });
    memcpy(y, helpers::getRawPointer(outputs), (size_t)N * L * sizeof(real_t));
}

REF_API void ref_ff_backward(int act, int P, int L, real_t bias, int N, const real_t *w, const real_t *x, const real_t *y,
                             real_t *outErr, real_t *prevErr, real_t *wu)
{
    rv weights(w, w + (size_t)L * (P + 1)), plOutputs(x, x + (size_t)N * P), outputs(y, y + (size_t)N * L);
    rv outputErrors(outErr, outErr + (size_t)N * L), plErrors((size_t)N * P), weightUpdates((size_t)L * (P + 1));
    by_act(act, [&](auto a) { ff_backward<decltype(a)>(P, L, bias, N, weights, plOutputs, outputs, outputErrors, prevErr ? &plErrors : 0, weightUpdates); });
    memcpy(outErr, helpers::getRawPointer(outputErrors), (size_t)N * L * sizeof(real_t));
    if (prevErr) memcpy(prevErr, helpers::getRawPointer(plErrors), (size_t)N * P * sizeof(real_t));
    memcpy(wu, helpers::getRawPointer(weightUpdates), (size_t)L * (P + 1) * sizeof(real_t));
}

// SoftmaxLayer<Cpu,Identity>::computeForwardPass, SoftmaxLayer.cu:250-315
REF_API void ref_softmax_forward(int P, int L, real_t bias, int N, const char *patTypes, const real_t *w, const real_t *x, real_t *y, real_t *patTmp)
{
    rv weights(w, w + (size_t)L * (P + 1)), plOutputs(x, x + (size_t)N * P), outputs(y, y + (size_t)N * L), m_patTmp(patTmp, patTmp + N);
    Cpu::pattype_vector pat(patTypes, patTypes + N);
    ff_forward<activation_functions::Identity>(P, L, bias, N, weights, plOutputs, outputs);
    {
        internal::CalculateOffsetFn fn;
        fn.layerSize = L; fn.outputs = helpers::getRawPointer(outputs); fn.patTypes = helpers::getRawPointer(pat);
        thrust::transform(thrust::counting_iterator<int>(0), thrust::counting_iterator<int>(0) + N, m_patTmp.begin(), fn);
    }
    {
        internal::CalculateExpFn fn;
        fn.layerSize = L; fn.offsets = helpers::getRawPointer(m_patTmp);
        const int n = N * L;
        thrust::for_each(thrust::make_zip_iterator(thrust::make_tuple(outputs.begin(), thrust::counting_iterator<int>(0))),
                         thrust::make_zip_iterator(thrust::make_tuple(outputs.begin() + n, thrust::counting_iterator<int>(0) + n)), fn);
    }
    {
        internal::SumUpOutputsFn fn;
        fn.layerSize = L; fn.outputs = helpers::getRawPointer(outputs);
        thrust::for_each(thrust::make_zip_iterator(thrust::make_tuple(m_patTmp.begin(), thrust::counting_iterator<int>(0))),
                         thrust::make_zip_iterator(thrust::make_tuple(m_patTmp.begin() + N, thrust::counting_iterator<int>(0) + N)), fn);
    }
    {
        internal::NormalizeOutputsFn fn;
        fn.layerSize = L; fn.normFacts = helpers::getRawPointer(m_patTmp);
        const int n = N * L;
        thrust::for_each(thrust::make_zip_iterator(thrust::make_tuple(outputs.begin(), thrust::counting_iterator<int>(0))),
                         thrust::make_zip_iterator(thrust::make_tuple(outputs.begin() + n, thrust::counting_iterator<int>(0) + n)), fn);
    }
    memcpy(y, helpers::getRawPointer(outputs), (size_t)N * L * sizeof(real_t));
    memcpy(patTmp, helpers::getRawPointer(m_patTmp), (size_t)N * sizeof(real_t));
}

// SoftmaxLayer<Cpu,Identity>::computeBackwardPass, SoftmaxLayer.cu:317-353
REF_API void ref_softmax_backward(int P, int L, real_t bias, int N, const char *patTypes, const real_t *w, const real_t *x, const real_t *y,
                                  real_t *outErr, real_t *prevErr, real_t *wu, real_t *patTmp)
{
    rv weights(w, w + (size_t)L * (P + 1)), plOutputs(x, x + (size_t)N * P), outputs(y, y + (size_t)N * L), m_patTmp(patTmp, patTmp + N);
    rv outputErrors(outErr, outErr + (size_t)N * L), plErrors((size_t)N * P), weightUpdates((size_t)L * (P + 1));
    Cpu::pattype_vector pat(patTypes, patTypes + N);
    {
        internal::CalculateErrorOffsetFn fn;
        fn.layerSize = L; fn.outputs = helpers::getRawPointer(outputs); fn.outputErrors = helpers::getRawPointer(outputErrors);
        fn.patTypes = helpers::getRawPointer(pat);
        thrust::transform(thrust::counting_iterator<int>(0), thrust::counting_iterator<int>(0) + N, m_patTmp.begin(), fn);
    }
    {
        internal::CalculateErrorsFn fn;
        fn.layerSize = L; fn.errorOffsets = helpers::getRawPointer(m_patTmp);
        const int n = N * L;
        thrust::for_each(thrust::make_zip_iterator(thrust::make_tuple(outputErrors.begin(), outputs.begin(), thrust::counting_iterator<int>(0))),
                         thrust::make_zip_iterator(thrust::make_tuple(outputErrors.begin() + n, outputs.begin() + n, thrust::counting_iterator<int>(0) + n)), fn);
    }
    ff_backward<activation_functions::Identity>(P, L, bias, N, weights, plOutputs, outputs, outputErrors, prevErr ? &plErrors : 0, weightUpdates);
    memcpy(outErr, helpers::getRawPointer(outputErrors), (size_t)N * L * sizeof(real_t));
    if (prevErr) memcpy(prevErr, helpers::getRawPointer(plErrors), (size_t)N * P * sizeof(real_t));
    memcpy(wu, helpers::getRawPointer(weightUpdates), (size_t)L * (P + 1) * sizeof(real_t));
    memcpy(patTmp, helpers::getRawPointer(m_patTmp), (size_t)N * sizeof(real_t));
}

// helpers::Matrix<Cpu>::assignProduct / addProduct (Matrix.cu:218-349), the three products the path uses
REF_API void ref_matmul(int kind, real_t *c, const real_t *a, int rowsA, int colsA, const real_t *b, int rowsB, int colsB, int add)
{
    rv va(a, a + (size_t)rowsA * colsA), vb(b, b + (size_t)rowsB * colsB);
    const int rowsC = kind == 1 ? colsA : rowsA, colsC = kind == 2 ? rowsB : colsB;
    rv vc(c, c + (size_t)rowsC * colsC);
    Mat A(&va, rowsA, colsA), B(&vb, rowsB, colsB), C(&vc, rowsC, colsC);
    const bool tA = kind == 1, tB = kind == 2;
    if (add) C.addProduct(A, tA, B, tB); else C.assignProduct(A, tA, B, tB);
    memcpy(c, helpers::getRawPointer(vc), (size_t)rowsC * colsC * sizeof(real_t));
}
