// Microbenchmark (GPU box only): what does the vendor library reach on the training step's plain GEMM shapes (fp32 in, fp32 out)?
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdio.h>

int main() {
    rocblas_handle h;
    rocblas_create_handle(&h);
    float *A, *B, *C;
    hipMalloc(&A, (size_t)5120 * 3072 * 4);
    hipMalloc(&B, (size_t)5120 * 3072 * 4);
    hipMalloc(&C, (size_t)5120 * 3072 * 4);
    hipMemset(A, 0, (size_t)5120 * 3072 * 4);
    hipMemset(B, 0, (size_t)5120 * 3072 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const float one = 1.f, zero = 0.f;
    struct Shape { const char* name; rocblas_operation ta, tb; int m, n, k, lda, ldb, ldc; };
    // row-major C[n1][n2] = sum_m A[m][n1] B[m][n2]  ==  column-major C^T[n2][n1] = B^T-as-colmajor ... (op N, op T)
    const Shape shapes[] = {
        {"dW_hh   tn 3072 x 1024 x 5120", rocblas_operation_none, rocblas_operation_transpose, 1024, 3072, 5120, 1024, 3072, 1024},
        {"dW_ih   tn 3072 x 496 x 5120", rocblas_operation_none, rocblas_operation_transpose, 496, 3072, 5120, 496, 3072, 496},
        {"gi      nt 5120 x 3072 x 496", rocblas_operation_transpose, rocblas_operation_none, 3072, 5120, 496, 496, 496, 3072},
        {"dX      nt 5120 x 496 x 3072", rocblas_operation_transpose, rocblas_operation_none, 496, 5120, 3072, 3072, 3072, 496},
    };
    for (const Shape& s : shapes) {
        for (int w = 0; w < 3; ++w) rocblas_sgemm(h, s.ta, s.tb, s.m, s.n, s.k, &one, A, s.lda, B, s.ldb, &zero, C, s.ldc);
        hipEventRecord(e0, 0);
        const int n = 10;
        for (int i = 0; i < n; ++i) rocblas_sgemm(h, s.ta, s.tb, s.m, s.n, s.k, &one, A, s.lda, B, s.ldb, &zero, C, s.ldc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.1f us  %6.1f TFLOP/s\n", s.name, 1e3 * ms / n, 2.0 * s.m * s.n * s.k / (1e9 * ms / n));
    }
    return 0;
}
