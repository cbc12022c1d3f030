/* oracle/shim: CBLAS prototypes (the image has libmkl_rt.so but no headers).  Enum values per the CBLAS standard. */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
enum CBLAS_ORDER { CblasRowMajor = 101, CblasColMajor = 102 };
enum CBLAS_TRANSPOSE { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 };
void cblas_sgemm(enum CBLAS_ORDER, enum CBLAS_TRANSPOSE, enum CBLAS_TRANSPOSE, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc);
void cblas_dgemm(enum CBLAS_ORDER, enum CBLAS_TRANSPOSE, enum CBLAS_TRANSPOSE, int M, int N, int K, double alpha, const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc);
void cblas_sgemv(enum CBLAS_ORDER, enum CBLAS_TRANSPOSE, int M, int N, float alpha, const float* A, int lda, const float* X, int incX, float beta, float* Y, int incY);
void cblas_dgemv(enum CBLAS_ORDER, enum CBLAS_TRANSPOSE, int M, int N, double alpha, const double* A, int lda, const double* X, int incX, double beta, double* Y, int incY);
void cblas_saxpy(int N, float alpha, const float* X, int incX, float* Y, int incY);
void cblas_daxpy(int N, double alpha, const double* X, int incX, double* Y, int incY);
void cblas_sscal(int N, float alpha, float* X, int incX);
void cblas_dscal(int N, double alpha, double* X, int incX);
void cblas_scopy(int N, const float* X, int incX, float* Y, int incY);
void cblas_dcopy(int N, const double* X, int incX, double* Y, int incY);
float cblas_sdot(int N, const float* X, int incX, const float* Y, int incY);
double cblas_ddot(int N, const double* X, int incX, const double* Y, int incY);
float cblas_sasum(int N, const float* X, int incX);
double cblas_dasum(int N, const double* X, int incX);
#ifdef __cplusplus
}
#endif
