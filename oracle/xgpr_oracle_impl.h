/* TEST INFRASTRUCTURE ONLY -- see xgpr_oracle.c for the header comment.
 *
 * This file is included twice by xgpr_oracle.c, once with T = float
 * (SUF = f32) and once with T = double (SUF = f64).  Every function is a
 * plain-C restatement of one function of the reference CPU path; the
 * reference file:line each one follows is given above it (paths relative
 * to /root/reference/src/xGPR/random_feature_generation/cpu_rf_gen/).
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* shared_fht_functions/hadamard_transforms.cpp:83-127 (singleVectorTransform).
 * Un-normalised, in-place, natural (Sylvester) order radix-2 FHT; stages run
 * h = 1, 2, 4, ... dim/2.  The reference unrolls h = 1, 2, 4; the arithmetic
 * (one add and one subtract per butterfly, no reassociation) is identical. */
static void FN(vec_fht)(T *x, int dim)
{
    for (int h = 1; h < dim; h <<= 1) {
        for (int i = 0; i < dim; i += (h << 1)) {
            for (int j = i; j < i + h; j++) {
                T y = x[j + h];
                x[j + h] = x[j] - y;
                x[j] = x[j] + y;
            }
        }
    }
}

/* SORF / SRHT normaliser, computed exactly as the reference does in type T:
 * shared_fht_functions/shared_rfgen_ops.cpp:54-55 and :23-24
 *     T norm_constant = log2(dim) / 2;  norm_constant = 1 / pow(2, norm_constant); */
static T FN(norm_constant)(int dim)
{
    T nc = (T)(log2((double)dim) / 2);
    nc = (T)(1 / pow(2.0, (double)nc));
    return nc;
}

/* shared_fht_functions/hadamard_transforms.cpp:17-71 (transformRows): every
 * contiguous block of dim2 elements of the [nrows, dim1, dim2] array is
 * transformed independently. */
void FN(orc_fht_rows)(T *x, long nrows, int dim1, int dim2)
{
    long nvec = nrows * (long)dim1;
    #pragma omp parallel for schedule(static)
    for (long v = 0; v < nvec; v++)
        FN(vec_fht)(x + v * dim2, dim2);
}

/* shared_fht_functions/shared_rfgen_ops.cpp:51-78 (singleVectorSORF):
 * three rounds of { x[i] *= radem[s,0,off+i] * norm ; FHT }. */
static void FN(vec_sorf)(T *buf, const int8_t *radem, int repeat_position,
                         int radem_shape2, int dim)
{
    T nc = FN(norm_constant)(dim);
    const int8_t *re = radem + repeat_position;
    for (int s = 0; s < 3; s++) {
        for (int i = 0; i < dim; i++)
            buf[i] *= re[i] * nc;
        FN(vec_fht)(buf, dim);
        re += radem_shape2;
    }
}

/* shared_fht_functions/shared_rfgen_ops.cpp:92-114 (singleVectorRBFPostProcess). */
static void FN(vec_rbf_post)(const T *xdata, const T *chi, double *out,
                             int dim2, int num_freqs, long row, int rep,
                             double scaling_term)
{
    int output_start = rep * dim2;
    int end_position = num_freqs < (rep + 1) * dim2 ? num_freqs : (rep + 1) * dim2;
    end_position -= output_start;
    const T *chi_in = chi + output_start;
    double *xout = out + 2 * (long)output_start + row * 2 * (long)num_freqs;
    for (int i = 0; i < end_position; i++) {
        T prod = xdata[i] * chi_in[i];
        *xout += TCOS(prod) * scaling_term;   /* cos/sin evaluated in T, widened by the product */
        xout++;
        *xout += TSIN(prod) * scaling_term;
        xout++;
    }
}

/* shared_fht_functions/shared_rfgen_ops.cpp:125-156 (singleVectorRBFPostGrad). */
static void FN(vec_rbf_post_grad)(const T *xdata, const T *chi, double *out,
                                  double *grad, double sigma, int dim2,
                                  int num_freqs, long row, int rep,
                                  double scaling_term)
{
    int output_start = rep * dim2;
    int end_position = num_freqs < (rep + 1) * dim2 ? num_freqs : (rep + 1) * dim2;
    end_position -= output_start;
    const T *chi_in = chi + output_start;
    double *xout = out + 2 * (long)output_start + row * 2 * (long)num_freqs;
    double *gout = grad + 2 * (long)output_start + row * 2 * (long)num_freqs;
    for (int i = 0; i < end_position; i++) {
        T grad_val = xdata[i] * chi_in[i];
        T prod_val = (T)(grad_val * sigma);            /* T*double -> double -> T */
        T cos_val = (T)(TCOS(prod_val) * scaling_term); /* rounded back to T */
        T sin_val = (T)(TSIN(prod_val) * scaling_term);
        *xout += cos_val; xout++;
        *xout += sin_val; xout++;
        *gout -= sin_val * grad_val; gout++;            /* product in T */
        *gout += cos_val * grad_val; gout++;
    }
}

/* padded width: rbf_ops.cpp:56-59 / rbf_convolution.cpp:60-64:
 * 2^ceil(log2(max(width, 2))). */
static int FN(padded_width)(long width)
{
    double e = width > 2 ? (double)width : 2.0;
    return (int)pow(2.0, ceil(log2(e)));
}

/* rbf_ops/rbf_ops.cpp:27-106 (rbfFeatureGen_).  Returns 0, or a negative
 * code where the reference throws std::runtime_error:
 *   -1 "no datapoints", -2 "last dim of output must be even number",
 *   -3 "incorrect number of rffs and or freqs." */
int FN(orc_rbf_feature_gen)(const T *x, double *out, const int8_t *radem,
                            const T *chi, long n, int d, long out_rows,
                            long num_rffs, long num_freqs, long radem_shape2,
                            int fit_intercept)
{
    if (n == 0 || out_rows != n) return -1;
    if (num_rffs < 2 || (num_rffs & 1) != 0) return -2;
    if (2 * num_freqs != num_rffs || num_freqs > radem_shape2) return -3;
    int P = FN(padded_width)(d);
    if (radem_shape2 % P != 0) return -3;

    /* rbf_ops.cpp:64-69: the constant is typed T in the reference. */
    T norm;
    if (fit_intercept) norm = (T)sqrt(1.0 / ((double)num_freqs - 0.5));
    else               norm = (T)sqrt(1.0 / (double)num_freqs);
    int reps = (int)((num_freqs + P - 1) / P);

    #pragma omp parallel
    {
        T *buf = (T *)malloc(sizeof(T) * (size_t)P);
        #pragma omp for schedule(static)
        for (long i = 0; i < n; i++) {
            int pos = 0;
            for (int k = 0; k < reps; k++) {
                for (int m = 0; m < d; m++) buf[m] = x[i * d + m];
                for (int m = d; m < P; m++) buf[m] = 0;
                FN(vec_sorf)(buf, radem, pos, (int)radem_shape2, P);
                FN(vec_rbf_post)(buf, chi, out, P, (int)num_freqs, i, k, (double)norm);
                pos += P;
            }
        }
        free(buf);
    }
    return 0;
}

/* rbf_ops/rbf_ops.cpp:136-221 (rbfGrad_): here the constant is a double
 * (rbf_ops.cpp:180), and the input is NOT pre-multiplied by sigma.
 * Extra code -4 "Wrong array sizes." */
int FN(orc_rbf_grad)(const T *x, double *out, double *grad, const int8_t *radem,
                     const T *chi, long n, int d, long out_rows, long num_rffs,
                     long grad_rows, long grad_cols, long num_freqs,
                     long radem_shape2, double sigma, int fit_intercept)
{
    if (n == 0 || out_rows != n) return -1;
    if (num_rffs < 2 || (num_rffs & 1) != 0) return -2;
    if (2 * num_freqs != num_rffs || num_freqs > radem_shape2) return -3;
    if (grad_rows != out_rows || grad_cols != num_rffs) return -4;
    int P = FN(padded_width)(d);
    if (radem_shape2 % P != 0) return -3;

    double norm;
    if (fit_intercept) norm = sqrt(1.0 / ((double)num_freqs - 0.5));
    else               norm = sqrt(1.0 / (double)num_freqs);
    int reps = (int)((num_freqs + P - 1) / P);

    #pragma omp parallel
    {
        T *buf = (T *)malloc(sizeof(T) * (size_t)P);
        #pragma omp for schedule(static)
        for (long i = 0; i < n; i++) {
            int pos = 0;
            for (int k = 0; k < reps; k++) {
                for (int m = 0; m < d; m++) buf[m] = x[i * d + m];
                for (int m = d; m < P; m++) buf[m] = 0;
                FN(vec_sorf)(buf, radem, pos, (int)radem_shape2, P);
                FN(vec_rbf_post_grad)(buf, chi, out, grad, sigma, P,
                                      (int)num_freqs, i, k, norm);
                pos += P;
            }
        }
        free(buf);
    }
    return 0;
}

/* rbf_ops/ard_ops.cpp:39-124 (ardGrad_): random features AND their gradient w.r.t. the
 * per-group lengthscales of the MiniARD kernel from a dense precomputed weight matrix
 * (num_freqs x d).  The product x[k] * w[j,k] is formed in T and widened (:100); the
 * group sums, the projection and cos/sin are double; the constant is typed T (:86-91).
 * The op only chooses the constant with fit_intercept -- column 0 is set by the Python
 * caller (kernel_baseclass.py:356-359).  Codes: -1 "no datapoints", -4 "Wrong array sizes.". */
int FN(orc_mini_ard_grad)(const T *x, double *out, const T *weights,
                          const int32_t *sigma_map, const double *sigma_vals,
                          double *grad, long n, long d, long out_rows, long num_rffs,
                          long num_freqs, long w_cols, long map_len, long sig_len,
                          long grad_rows, long grad_cols, long num_lengthscales,
                          int fit_intercept)
{
    if (n == 0 || out_rows != n) return -1;
    if (grad_rows != out_rows || grad_cols != num_rffs) return -4;
    if (w_cols != d) return -4;
    if (num_rffs != 2 * num_freqs || map_len != w_cols) return -4;
    if (sig_len != map_len) return -4;
    T norm;
    if (fit_intercept) norm = (T)sqrt(1.0 / ((double)num_freqs - 0.5));
    else               norm = (T)sqrt(1.0 / (double)num_freqs);
    long nl = num_lengthscales;
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        const T *xi = x + i * d;
        for (long j = 0; j < num_freqs; j++) {
            double *g = grad + (i * num_rffs + 2 * j) * nl;
            const T *w = weights + j * d;
            double rf = 0;
            for (long k = 0; k < d; k++) {
                double dot = xi[k] * w[k];
                g[sigma_map[k]] += dot;
                rf += sigma_vals[k] * dot;
            }
            double c = norm * cos(rf), sn = norm * sin(rf);
            out[i * num_rffs + 2 * j] = c;
            out[i * num_rffs + 2 * j + 1] = sn;
            for (long k = 0; k < nl; k++) {
                double gv = g[k];
                g[k] = -gv * sn;
                g[k + nl] = gv * c;
            }
        }
    }
    return 0;
}

/* Sequence-length validation shared by the conv ops:
 * convolution_ops/rbf_convolution.cpp:55-82, conv1d_operations.cpp:52-81.
 *   -5 "wrong array sizes", -6 "invalid conv_width",
 *   -7 "All sequence lengths must be >= conv width and < array size." */
static int FN(check_seqlens)(const int32_t *seqlen, long nseq, long n, int L,
                             int conv_width)
{
    if (nseq != n) return -5;
    if (L < conv_width || conv_width <= 0) return -6;
    int32_t mn = 2147483647, mx = 0;
    for (long i = 0; i < nseq; i++) {
        if (seqlen[i] > mx) mx = seqlen[i];
        if (seqlen[i] < mn) mn = seqlen[i];
    }
    if (mx > L || mn < conv_width) return -7;
    return 0;
}

/* convolution_ops/rbf_convolution.cpp:23-140 (convRBFFeatureGen_).
 * scaling_type: 0 none, 1 sqrt, 2 full (rbf_convolution.h:21-23). */
int FN(orc_conv1d_fgen)(const T *x, double *out, const int8_t *radem,
                        const T *chi, const int32_t *seqlen, long n, int L,
                        int C, long out_rows, long num_rffs, long num_freqs,
                        long radem_shape2, long nseq, int conv_width,
                        int scaling_type)
{
    if (n == 0 || out_rows != n) return -1;
    if (num_rffs < 2 || (num_rffs & 1) != 0) return -2;
    if (2 * num_freqs != num_rffs || num_freqs > radem_shape2) return -3;
    if (nseq != n) return -5;
    if (L < conv_width || conv_width <= 0) return -6;
    int P = FN(padded_width)((long)conv_width * C);
    if (radem_shape2 % P != 0) return -3;
    int rc = FN(check_seqlens)(seqlen, nseq, n, L, conv_width);
    if (rc) return rc;

    double scaling_term = sqrt(1.0 / (double)num_freqs);
    int reps = (int)((num_freqs + P - 1) / P);
    int win = conv_width * C;

    #pragma omp parallel
    {
        T *buf = (T *)malloc(sizeof(T) * (size_t)P);
        #pragma omp for schedule(dynamic, 1)
        for (long i = 0; i < n; i++) {
            int nkmers = seqlen[i] - conv_width + 1;
            double row_scaler;
            if (scaling_type == 1)      row_scaler = scaling_term / sqrt((double)nkmers);
            else if (scaling_type == 2) row_scaler = scaling_term / (double)nkmers;
            else                        row_scaler = scaling_term;
            for (int j = 0; j < nkmers; j++) {
                const T *xe = x + i * (long)L * C + (long)j * C;
                int pos = 0;
                for (int k = 0; k < reps; k++) {
                    for (int m = 0; m < win; m++) buf[m] = xe[m];
                    for (int m = win; m < P; m++) buf[m] = 0;
                    FN(vec_sorf)(buf, radem, pos, (int)radem_shape2, P);
                    FN(vec_rbf_post)(buf, chi, out, P, (int)num_freqs, i, k, row_scaler);
                    pos += P;
                }
            }
        }
        free(buf);
    }
    return 0;
}

/* convolution_ops/rbf_convolution.cpp:160-282 (convRBFGrad_): as conv1d_fgen
 * with singleVectorRBFPostGrad; the input is not pre-multiplied by sigma. */
int FN(orc_conv_grad)(const T *x, double *out, double *grad, const int8_t *radem,
                      const T *chi, const int32_t *seqlen, long n, int L, int C,
                      long out_rows, long num_rffs, long grad_rows, long grad_cols,
                      long num_freqs, long radem_shape2, long nseq, double sigma,
                      int conv_width, int scaling_type)
{
    if (n == 0 || out_rows != n) return -1;
    if (num_rffs < 2 || (num_rffs & 1) != 0) return -2;
    if (2 * num_freqs != num_rffs || num_freqs > radem_shape2) return -3;
    if (grad_rows != out_rows || grad_cols != num_rffs) return -4;
    if (nseq != n) return -5;
    if (L < conv_width || conv_width <= 0) return -6;
    int P = FN(padded_width)((long)conv_width * C);
    if (radem_shape2 % P != 0) return -3;
    int rc = FN(check_seqlens)(seqlen, nseq, n, L, conv_width);
    if (rc) return rc;

    double scaling_term = sqrt(1.0 / (double)num_freqs);
    int reps = (int)((num_freqs + P - 1) / P);
    int win = conv_width * C;

    #pragma omp parallel
    {
        T *buf = (T *)malloc(sizeof(T) * (size_t)P);
        #pragma omp for schedule(dynamic, 1)
        for (long i = 0; i < n; i++) {
            int nkmers = seqlen[i] - conv_width + 1;
            double row_scaler;
            if (scaling_type == 1)      row_scaler = scaling_term / sqrt((double)nkmers);
            else if (scaling_type == 2) row_scaler = scaling_term / (double)nkmers;
            else                        row_scaler = scaling_term;
            for (int j = 0; j < nkmers; j++) {
                const T *xe = x + i * (long)L * C + (long)j * C;
                int pos = 0;
                for (int k = 0; k < reps; k++) {
                    for (int m = 0; m < win; m++) buf[m] = xe[m];
                    for (int m = win; m < P; m++) buf[m] = 0;
                    FN(vec_sorf)(buf, radem, pos, (int)radem_shape2, P);
                    FN(vec_rbf_post_grad)(buf, chi, out, grad, sigma, P,
                                          (int)num_freqs, i, k, row_scaler);
                    pos += P;
                }
            }
        }
        free(buf);
    }
    return 0;
}

/* convolution_ops/conv1d_operations.cpp:23-124 (conv1dMaxpoolFeatureGen_) and
 * :146-168 (singleVectorMaxpoolPostProcess): out = max(out, chi * sorf(x)),
 * float output, num_freqs == num_rffs, radem_shape2 == reps * P exactly. */
int FN(orc_conv1d_maxpool)(const T *x, float *out, const int8_t *radem,
                           const T *chi, const int32_t *seqlen, long n, int L,
                           int C, long out_rows, long num_rffs, long num_freqs,
                           long radem_shape2, long nseq, int conv_width)
{
    if (n == 0 || out_rows != n) return -1;
    if (num_rffs < 2 || (num_rffs & 1) != 0) return -2;
    if (num_freqs != num_rffs || num_freqs > radem_shape2) return -3;
    if (nseq != n) return -5;
    if (L < conv_width || conv_width <= 0) return -6;
    int P = FN(padded_width)((long)conv_width * C);
    int reps = (int)((num_freqs + P - 1) / P);
    if (radem_shape2 % P != 0 || radem_shape2 != (long)reps * P) return -3;
    int rc = FN(check_seqlens)(seqlen, nseq, n, L, conv_width);
    if (rc) return rc;
    int win = conv_width * C;

    #pragma omp parallel
    {
        T *buf = (T *)malloc(sizeof(T) * (size_t)P);
        #pragma omp for schedule(dynamic, 1)
        for (long i = 0; i < n; i++) {
            int nkmers = seqlen[i] - conv_width + 1;
            for (int j = 0; j < nkmers; j++) {
                const T *xe = x + i * (long)L * C + (long)j * C;
                int pos = 0;
                for (int k = 0; k < reps; k++) {
                    for (int m = 0; m < win; m++) buf[m] = xe[m];
                    for (int m = win; m < P; m++) buf[m] = 0;
                    FN(vec_sorf)(buf, radem, pos, (int)radem_shape2, P);
                    int output_start = k * P;
                    int endp = (int)(num_freqs < (long)(k + 1) * P ? num_freqs : (long)(k + 1) * P);
                    endp -= output_start;
                    float *xo = out + output_start + i * num_freqs;
                    for (int q = 0; q < endp; q++) {
                        float prod = (float)(buf[q] * chi[output_start + q]);
                        xo[q] = xo[q] > prod ? xo[q] : prod;   /* std::max(*xOut, prodVal) */
                    }
                    pos += P;
                }
            }
        }
        free(buf);
    }
    return 0;
}

/* basic_ops/transform_functions.cpp:95-121 (SRHTBlockTransform) with
 * shared_rfgen_ops.cpp:19-34 (multiplyByDiagonalRademacherMat2D):
 * X <- FHT(X * radem * norm) over rows.
 *   -1 "no datapoints", -8 "incorrect array dims passed",
 *   -9 "last dim not power of 2 > 1" / "last dim not power of 2". */
int FN(orc_srht)(T *x, const int8_t *radem, long n, long dim, long radem_len)
{
    if (n == 0) return -1;
    if (dim != radem_len) return -8;
    if (dim < 2) return -9;
    if ((dim & (dim - 1)) != 0) return -9;
    T nc = FN(norm_constant)((int)dim);
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        T *xe = x + i * dim;
        for (long j = 0; j < dim; j++)
            xe[j] *= radem[j] * nc;
        FN(vec_fht)(xe, (int)dim);
    }
    return 0;
}

/* basic_ops/transform_functions.cpp:59-80 (fastHadamard2dArray_) and :22-46
 * (fastHadamard3dArray_): validated bare FHT over the last axis. */
int FN(orc_fht)(T *x, long n, long dim1, long dim2)
{
    if (n == 0) return -1;
    if (dim2 < 2) return -9;
    if ((dim2 & (dim2 - 1)) != 0) return -9;
    FN(orc_fht_rows)(x, n, (int)dim1, (int)dim2);
    return 0;
}

#undef CAT_
#undef CAT
#undef FN
