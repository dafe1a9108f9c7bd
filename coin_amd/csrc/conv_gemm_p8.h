// Internal interface between conv_gemm.hip (the C-ABI entry points) and conv_gemm_p8.hip (the persistent 256x256x64 cores).
#pragma once
#include <hip/hip_runtime.h>

#define COIN_HIDDEN __attribute__((visibility("hidden")))

COIN_HIDDEN bool coin_p8_nt_ok(int M, int N, int K, int mode, int Cin, int lda, int ldb);
COIN_HIDDEN size_t coin_p8_nt_workspace_bytes(int M, int N, int K);
COIN_HIDDEN int coin_p8_nt_launch(const void* A, int lda, int mode, int H, int W, int Cin, const void* B, int ldb, void* C, int ldc, const void* R,
                                  int ldr, int M, int N, int K, float* stats, long long stats_rows, void* workspace, size_t workspace_bytes,
                                  hipStream_t st, int rp_h = 0, int rp_w = 0);
COIN_HIDDEN bool coin_p8_tn_ok(int M, int Cout, int Cin, int Ktot, int mode);
COIN_HIDDEN size_t coin_p8_tn_workspace_bytes(int M, int Cout, int Ktot);
COIN_HIDDEN int coin_p8_tn_launch(const void* GY, const void* X, int mode, int H, int W, int Cin, int M, int Cout, int Ktot, float* dW, void* workspace,
                                  hipStream_t st);
// 128 x 128 x 32 core for the small maps (conv_gemm_s4.hip): four workgroups per CU; statistics partials per 128-row tile
COIN_HIDDEN bool coin_s4_nt_ok(int M, int N, int K, int mode, int Cin, int lda, int ldb);
COIN_HIDDEN bool coin_s4_nt_wanted(int M, int N, int K);
COIN_HIDDEN size_t coin_s4_nt_workspace_bytes(int M, int N, int K);
COIN_HIDDEN int coin_s4_nt_launch(const void* A, int lda, int mode, int H, int W, int Cin, const void* B, int ldb, void* C, int ldc, const void* R,
                                  int ldr, int M, int N, int K, float* stats, long long stats_rows, void* workspace, size_t workspace_bytes,
                                  hipStream_t st, int rp_h = 0, int rp_w = 0);
COIN_HIDDEN bool coin_s4_tn_ok(int M, int Cout, int Cin, int Ktot, int mode);
COIN_HIDDEN bool coin_s4_tn_wanted(int M, int Cout, int Cin);
COIN_HIDDEN size_t coin_s4_tn_workspace_bytes(int M, int Cout, int Ktot);
COIN_HIDDEN int coin_s4_tn_launch(const void* GY, const void* X, int mode, int H, int W, int Cin, int M, int Cout, int Ktot, float* dW, void* workspace,
                                  hipStream_t st);
#ifdef COIN_LAB   // development switches of tools/gemm_lab: not part of the product library (built without -DCOIN_LAB)
COIN_HIDDEN extern int coin_conv_gemm_force_impl;
COIN_HIDDEN extern int coin_p8_debug;
COIN_HIDDEN int coin_p8_read_stamps(long long* out, int n);
COIN_HIDDEN extern int coin_p8_splitk;
COIN_HIDDEN extern int coin_p8_stagger;
COIN_HIDDEN extern int coin_s4_split;
COIN_HIDDEN extern int coin_s4_stages;
COIN_HIDDEN extern int coin_s4_maxwg;
COIN_HIDDEN extern int coin_s4_debug;
COIN_HIDDEN extern int coin_s4_tn_wpc;
#endif
