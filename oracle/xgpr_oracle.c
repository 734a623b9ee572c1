/* xgpr_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference CPU algorithm for the hot path
 * (SORF / fast-Hadamard random-feature generation) of jlparkI/xGPR
 * v0.4.9, written from the reference's behaviour -- each function cites
 * the reference file:line it follows (see xgpr_oracle_impl.h).
 *
 * Who may use it: tests/, __graft_entry__.smoke() and the `cpu_baseline`
 * leg of bench.py -- as the checker / the timed CPU baseline, never as a
 * fallback for the HIP path.  Nothing under xgpr_amd/ imports or links it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every entry
 * point bit-for-bit against the reference's own compiled arithmetic core
 * (oracle/_ref, built from /root/reference by oracle/Makefile, only in the
 * authoring container) and tests/test_oracle_golden.py checks it against
 * the golden vectors under tests/golden/ that the reference produced.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off; no
 * -march=native so that no FMA contraction can change a rounding -- the
 * reference is built without -march flags too, CMakeLists.txt).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* float instantiation: cos/sin resolve to the float overloads in the
 * reference (C++ <math.h>), i.e. glibc cosf/sinf. */
#define T float
#define SUF f32
#define TCOS(v) cosf(v)
#define TSIN(v) sinf(v)
#include "xgpr_oracle_impl.h"
#undef T
#undef SUF
#undef TCOS
#undef TSIN

#define T double
#define SUF f64
#define TCOS(v) cos(v)
#define TSIN(v) sin(v)
#include "xgpr_oracle_impl.h"
#undef T
#undef SUF
#undef TCOS
#undef TSIN

/* Threads the OpenMP team will use (reported next to every CPU timing). */
int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* w += Z^T (Z v) for one chunk, Z row-major [n, M] f64, v,w [M] -- the dense
 * step of src/xGPR/fitting_toolkit/cg_tools.py:189-191 for k = 1, used only by
 * bench.py's cpu_baseline leg (numpy/BLAS does the same job in the tests). */
void orc_ztz_matvec_f64(const double *z, const double *v, double *w, long n, long M)
{
    #pragma omp parallel
    {
        double *wl = (double *)calloc((size_t)M, sizeof(double));
        #pragma omp for schedule(static)
        for (long i = 0; i < n; i++) {
            const double *zr = z + i * M;
            double t = 0.0;
            for (long j = 0; j < M; j++) t += zr[j] * v[j];
            for (long j = 0; j < M; j++) wl[j] += zr[j] * t;
        }
        #pragma omp critical
        for (long j = 0; j < M; j++) w[j] += wl[j];
        free(wl);
    }
}
