// Drives the functors of the reference's layers/MulticlassClassificationLayer.cu (see ref_common.h).
// Signatures equal orc_mcc_* of oracle/currennt_oracle.c.
#include <thrust/transform_reduce.h>      // (the reference .cu uses thrust::transform_reduce without including this header)
#include "/root/reference/currennt_lib/src/layers/MulticlassClassificationLayer.cu"
#include "ref_common.h"

// MulticlassClassificationLayer<Cpu>::calculateError, .cu:194-213
REF_API real_t ref_mcc_error(int L, int N, const int *targetClasses, const real_t *outputs)
{
    Cpu::int_vector m_patTargetClasses(targetClasses, targetClasses + N);
    internal::ComputeCrossEntropyErrorFn fn;
    fn.layerSize = L; fn.outputs = outputs;
    const int n = N;
    real_t error = thrust::transform_reduce(
        thrust::make_zip_iterator(thrust::make_tuple(m_patTargetClasses.begin(), thrust::counting_iterator<int>(0))),
        thrust::make_zip_iterator(thrust::make_tuple(m_patTargetClasses.begin() + n, thrust::counting_iterator<int>(0) + n)),
        fn, (real_t)0, thrust::plus<real_t>());
    return -error;
}

// MulticlassClassificationLayer<Cpu>::countCorrectClassifications, .cu:159-177
REF_API int ref_mcc_correct(int L, int N, const int *targetClasses, const real_t *outputs)
{
    Cpu::int_vector m_patTargetClasses(targetClasses, targetClasses + N);
    internal::CountCorrectClassificationsFn fn;
    fn.layerSize = L; fn.outputs = outputs;
    const int n = N;
    return thrust::transform_reduce(
        thrust::make_zip_iterator(thrust::make_tuple(m_patTargetClasses.begin(), thrust::counting_iterator<int>(0))),
        thrust::make_zip_iterator(thrust::make_tuple(m_patTargetClasses.begin() + n, thrust::counting_iterator<int>(0) + n)),
        fn, 0, thrust::plus<int>());
}

// MulticlassClassificationLayer<Cpu>::computeBackwardPass, .cu:220-240
REF_API void ref_mcc_backward(int L, int N, const int *targetClasses, const real_t *outputs, real_t *outErr)
{
    Cpu::int_vector m_patTargetClasses(targetClasses, targetClasses + N);
    Cpu::real_vector outputErrors((size_t)N * L);
    thrust::fill_n(outputErrors.begin(), N * L, (real_t)0);
    internal::ComputeOutputErrorFn fn;
    fn.layerSize = L; fn.outputs = outputs; fn.outputErrors = helpers::getRawPointer(outputErrors);
    const int n = N;
    thrust::for_each(thrust::make_zip_iterator(thrust::make_tuple(m_patTargetClasses.begin(), thrust::counting_iterator<int>(0))),
                     thrust::make_zip_iterator(thrust::make_tuple(m_patTargetClasses.begin() + n, thrust::counting_iterator<int>(0) + n)), fn);
    memcpy(outErr, helpers::getRawPointer(outputErrors), (size_t)N * L * sizeof(real_t));
}
