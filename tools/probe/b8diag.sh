# diagnostic builds of the 8-phase GEMM (results wrong, timing only): tools/probe/gemm_bench_<x>
for v in NOSTAGE NOMFMA NOEPI; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DB8_DIAG_$v -c lstm-rnn_amd/csrc/cn_gemm_big.hip -o /tmp/big_$v.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/gb.o lstm-rnn_amd/csrc/cn_gemm.o lstm-rnn_amd/csrc/cn_gemm_tn_big.o lstm-rnn_amd/csrc/cn_gemm_nt_mid.o /tmp/big_$v.o -o tools/probe/gemm_bench_$v
done
