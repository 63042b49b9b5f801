/*
 * airwave_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See airwave_oracle.h for the scope and the parity-pin statement.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference repository root).  Arithmetic is float32 wherever the reference's is
 * (vDSP single precision); Apple's vDSP itself is closed source, so the FFT here
 * is an ordinary radix-2 transform that follows vDSP's documented *conventions*
 * (packed DC/Nyquist, forward = 2 x DFT, inverse unnormalised), which is what the
 * reference's 0.25/fftSize scale (ConvolutionEngine.swift:356) depends on.
 */
#include "airwave_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * vDSP stand-ins (call-site inventory: SURVEY.md §2.1)
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    int log2n;      /* of the REAL transform length N */
    int n;          /* N */
    int half;       /* N/2: length of the complex transform */
    float *tw_re;   /* e^{-2 pi i k / half}, k < half/2 : complex FFT twiddles */
    float *tw_im;
    float *rw_re;   /* e^{-2 pi i k / N}, k < half : real-FFT split twiddles */
    float *rw_im;
    int *bitrev;    /* bit reversal for length half */
} orc_fftsetup;

/* vDSP_create_fftsetup(log2n, kFFTRadix2)  — ConvolutionEngine.swift:82,
 * FFTSetupManager.swift:52 */
static orc_fftsetup *fftsetup_create(int log2n) {
    if (log2n < 2 || log2n > 24) return NULL;
    orc_fftsetup *s = (orc_fftsetup *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->log2n = log2n;
    s->n = 1 << log2n;
    s->half = s->n / 2;
    int q = s->half / 2 > 0 ? s->half / 2 : 1;
    s->tw_re = (float *)malloc(sizeof(float) * (size_t)q);
    s->tw_im = (float *)malloc(sizeof(float) * (size_t)q);
    s->rw_re = (float *)malloc(sizeof(float) * (size_t)s->half);
    s->rw_im = (float *)malloc(sizeof(float) * (size_t)s->half);
    s->bitrev = (int *)malloc(sizeof(int) * (size_t)s->half);
    if (!s->tw_re || !s->tw_im || !s->rw_re || !s->rw_im || !s->bitrev) return NULL;
    for (int k = 0; k < q; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)s->half;
        s->tw_re[k] = (float)cos(a);
        s->tw_im[k] = (float)sin(a);
    }
    for (int k = 0; k < s->half; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)s->n;
        s->rw_re[k] = (float)cos(a);
        s->rw_im[k] = (float)sin(a);
    }
    int bits = log2n - 1;
    for (int i = 0; i < s->half; ++i) {
        int r = 0;
        for (int b = 0; b < bits; ++b)
            if (i & (1 << b)) r |= 1 << (bits - 1 - b);
        s->bitrev[i] = r;
    }
    return s;
}

static void fftsetup_destroy(orc_fftsetup *s) {
    if (!s) return;
    free(s->tw_re); free(s->tw_im); free(s->rw_re); free(s->rw_im); free(s->bitrev);
    free(s);
}

/* In-place radix-2 complex FFT of length s->half on split arrays.
 * sign = -1 forward (e^{-i}), +1 inverse (e^{+i}); unnormalised both ways. */
static void cfft_split(const orc_fftsetup *s, float *re, float *im, int sign) {
    const int n = s->half;
    for (int i = 0; i < n; ++i) {
        int j = s->bitrev[i];
        if (j > i) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int hl = len >> 1;
        const int step = n / len;
        for (int base = 0; base < n; base += len) {
            for (int k = 0; k < hl; ++k) {
                const float wr = s->tw_re[k * step];
                const float wi = sign < 0 ? s->tw_im[k * step] : -s->tw_im[k * step];
                const int a = base + k, b = a + hl;
                const float xr = re[b] * wr - im[b] * wi;
                const float xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] = re[a] + xr; im[a] = im[a] + xi;
            }
        }
    }
}

/* vDSP_ctoz(stride 2 -> 1): even samples -> realp, odd -> imagp.
 * ConvolutionEngine.swift:169,248 */
static void vdsp_ctoz(const float *interleaved, float *realp, float *imagp, int half) {
    for (int i = 0; i < half; ++i) {
        realp[i] = interleaved[2 * i];
        imagp[i] = interleaved[2 * i + 1];
    }
}

/* vDSP_ztoc: ConvolutionEngine.swift:362 */
static void vdsp_ztoc(const float *realp, const float *imagp, float *interleaved, int half) {
    for (int i = 0; i < half; ++i) {
        interleaved[2 * i] = realp[i];
        interleaved[2 * i + 1] = imagp[i];
    }
}

/* vDSP_fft_zrip forward (ConvolutionEngine.swift:174,252): in-place real FFT in the
 * packed format: realp[0] = DC, imagp[0] = Nyquist, bins 1..N/2-1 in (realp[k], imagp[k]);
 * every value is 2 x the mathematical DFT. */
static void vdsp_fft_zrip_forward(const orc_fftsetup *s, float *realp, float *imagp) {
    const int h = s->half;
    cfft_split(s, realp, imagp, -1);
    const float z0r = realp[0], z0i = imagp[0];
    realp[0] = 2.0f * (z0r + z0i);
    imagp[0] = 2.0f * (z0r - z0i);
    for (int k = 1; k <= h / 2; ++k) {
        const int m = h - k;
        const float ar = realp[k], ai = imagp[k];
        const float br = realp[m], bi = imagp[m];
        /* E2 = Z[k] + conj(Z[m]) ; D = Z[k] - conj(Z[m]) */
        const float e_r = ar + br, e_i = ai - bi;
        const float d_r = ar - br, d_i = ai + bi;
        /* 2X[k] = E2 - i W^k D */
        const float wr = s->rw_re[k], wi = s->rw_im[k];
        const float t_r = wr * d_r - wi * d_i;   /* W^k D */
        const float t_i = wr * d_i + wi * d_r;
        const float xk_r = e_r + t_i, xk_i = e_i - t_r;
        /* 2X[m] with m = h-k: E2' = conj(E2), D' = -conj(D), W^m = -conj(W^k)
         *  => 2X[m] = conj(E2) - i * conj(W^k D) */
        const float xm_r = e_r - t_i, xm_i = -e_i - t_r;
        realp[k] = xk_r; imagp[k] = xk_i;
        if (m != k) { realp[m] = xm_r; imagp[m] = xm_i; }
    }
}

/* vDSP_fft_zrip inverse (ConvolutionEngine.swift:353): packed half spectrum S ->
 * packed real signal x[n] = sum_{k<N} S_full[k] e^{+2 pi i n k / N} (unnormalised). */
static void vdsp_fft_zrip_inverse(const orc_fftsetup *s, float *realp, float *imagp) {
    const int h = s->half;
    const float s0 = realp[0], sh = imagp[0];
    realp[0] = s0 + sh;     /* Fe[0] + i Fo[0] */
    imagp[0] = s0 - sh;
    for (int k = 1; k <= h / 2; ++k) {
        const int m = h - k;
        const float ar = realp[k], ai = imagp[k];
        const float br = realp[m], bi = imagp[m];
        /* Fe[k] = S[k] + conj(S[m]);  Fo[k] = (S[k] - conj(S[m])) * conj(W^k) */
        const float fe_r = ar + br, fe_i = ai - bi;
        const float d_r = ar - br, d_i = ai + bi;
        const float wr = s->rw_re[k], wi = -s->rw_im[k];
        const float fo_r = wr * d_r - wi * d_i;
        const float fo_i = wr * d_i + wi * d_r;
        /* Z[k] = Fe + i Fo */
        const float zk_r = fe_r - fo_i, zk_i = fe_i + fo_r;
        /* index m: Fe[m] = conj(Fe[k]), Fo[m] = conj(Fo[k]) (since W^-m = -conj(W^-k), D' = -conj D) */
        const float zm_r = fe_r + fo_i, zm_i = -fe_i + fo_r;
        realp[k] = zk_r; imagp[k] = zk_i;
        if (m != k) { realp[m] = zm_r; imagp[m] = zm_i; }
    }
    cfft_split(s, realp, imagp, +1);
}

/* vDSP_zvmul(A, B, C, n, conjugate = +1): C = A * B  (ConvolutionEngine.swift:311,346) */
static void vdsp_zvmul(const float *ar, const float *ai, const float *br, const float *bi, float *cr,
                       float *ci, int n) {
    for (int i = 0; i < n; ++i) {
        const float r = ar[i] * br[i] - ai[i] * bi[i];
        const float m = ar[i] * bi[i] + ai[i] * br[i];
        cr[i] = r; ci[i] = m;
    }
}

/* vDSP_zvadd (ConvolutionEngine.swift:347) */
static void vdsp_zvadd(const float *ar, const float *ai, float *cr, float *ci, int n) {
    for (int i = 0; i < n; ++i) { cr[i] += ar[i]; ci[i] += ai[i]; }
}

/* ------------------------------------------------------------------------------------------
 * ConvolutionEngine
 * ---------------------------------------------------------------------------------------- */
struct orc_engine {
    int log2n, fft_size, fft_half, block_size;
    int partition_count, partition_count_pow2;
    orc_fftsetup *setup;
    float *input_buffer;          /* fftSize    (:102) */
    float *input_overlap;         /* blockSize  (:105) */
    float *fdl_re, *fdl_im;       /* [P_pow2][fftHalf]  (:129-130) */
    float *hrir_re, *hrir_im;     /* [P_pow2][fftHalf]  (:131-132) */
    float *split_re, *split_im;   /* (:109-111) */
    float *acc_re, *acc_im;       /* (:113-115) */
    float *tmp_re, *tmp_im;       /* (:117-119) */
    float *temp_output;           /* (:122) */
    int fdl_index;                /* (:39) */
};

static int is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

orc_engine *orc_engine_create(const float *hrir, int count, int block_size) {
    if (!hrir || count <= 0 || block_size < 2 || !is_pow2(block_size)) return NULL;
    orc_engine *e = (orc_engine *)calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->block_size = block_size;
    e->fft_size = block_size * 2;                        /* :72 */
    e->fft_half = e->fft_size / 2;                       /* :73 */
    int l2 = 0; while ((1 << l2) < e->fft_size) ++l2;    /* :74 */
    e->log2n = l2;
    e->setup = fftsetup_create(l2);                      /* :82 */
    if (!e->setup) { free(e); return NULL; }
    e->partition_count = (count + block_size - 1) / block_size;   /* :93 */
    int p2 = 1; while (p2 < e->partition_count) p2 <<= 1;         /* :96 */
    e->partition_count_pow2 = p2;

    const size_t h = (size_t)e->fft_half;
    e->input_buffer = (float *)calloc((size_t)e->fft_size, sizeof(float));
    e->input_overlap = (float *)calloc((size_t)block_size, sizeof(float));
    e->split_re = (float *)calloc(h, sizeof(float));
    e->split_im = (float *)calloc(h, sizeof(float));
    e->acc_re = (float *)calloc(h, sizeof(float));
    e->acc_im = (float *)calloc(h, sizeof(float));
    e->tmp_re = (float *)calloc(h, sizeof(float));
    e->tmp_im = (float *)calloc(h, sizeof(float));
    e->temp_output = (float *)calloc((size_t)block_size, sizeof(float));
    const size_t tot = (size_t)p2 * h;                   /* :126-138 */
    e->fdl_re = (float *)calloc(tot, sizeof(float));
    e->fdl_im = (float *)calloc(tot, sizeof(float));
    e->hrir_re = (float *)calloc(tot, sizeof(float));
    e->hrir_im = (float *)calloc(tot, sizeof(float));
    if (!e->input_buffer || !e->input_overlap || !e->split_re || !e->split_im || !e->acc_re ||
        !e->acc_im || !e->tmp_re || !e->tmp_im || !e->temp_output || !e->fdl_re || !e->fdl_im ||
        !e->hrir_re || !e->hrir_im) {
        orc_engine_destroy(e);
        return NULL;
    }

    /* 6. Process HRIR partitions  (:141-175) */
    float *pad = (float *)calloc((size_t)e->fft_size, sizeof(float));
    if (!pad) { orc_engine_destroy(e); return NULL; }
    for (int p = 0; p < e->partition_count; ++p) {
        memset(pad, 0, sizeof(float) * (size_t)e->fft_size);
        const int start = p * block_size;
        int end = start + block_size; if (end > count) end = count;
        for (int i = 0; i < end - start; ++i) pad[i] = hrir[start + i];
        float *hr = e->hrir_re + (size_t)p * h, *hi = e->hrir_im + (size_t)p * h;
        vdsp_ctoz(pad, hr, hi, e->fft_half);             /* :169 */
        vdsp_fft_zrip_forward(e->setup, hr, hi);         /* :174 */
    }
    free(pad);
    return e;
}

void orc_engine_destroy(orc_engine *e) {
    if (!e) return;
    fftsetup_destroy(e->setup);
    free(e->input_buffer); free(e->input_overlap); free(e->split_re); free(e->split_im);
    free(e->acc_re); free(e->acc_im); free(e->tmp_re); free(e->tmp_im); free(e->temp_output);
    free(e->fdl_re); free(e->fdl_im); free(e->hrir_re); free(e->hrir_im);
    free(e);
}

int orc_engine_block_size(const orc_engine *e) { return e->block_size; }
int orc_engine_partition_count(const orc_engine *e) { return e->partition_count; }

void orc_engine_process(orc_engine *e, const float *input, float *output) {
    const int B = e->block_size, H = e->fft_half, P = e->partition_count;
    /* 1. overlap-save input [previous | current]  (:237-243) */
    memcpy(e->input_buffer, e->input_overlap, sizeof(float) * (size_t)B);
    memcpy(e->input_buffer + B, input, sizeof(float) * (size_t)B);
    memcpy(e->input_overlap, input, sizeof(float) * (size_t)B);
    /* 2. FFT input  (:247-252) */
    vdsp_ctoz(e->input_buffer, e->split_re, e->split_im, H);
    vdsp_fft_zrip_forward(e->setup, e->split_re, e->split_im);
    /* 3. FDL update: index decremented modulo partitionCount  (:256-264) */
    e->fdl_index -= 1;
    if (e->fdl_index < 0) e->fdl_index += P;
    memcpy(e->fdl_re + (size_t)e->fdl_index * H, e->split_re, sizeof(float) * (size_t)H);
    memcpy(e->fdl_im + (size_t)e->fdl_index * H, e->split_im, sizeof(float) * (size_t)H);
    /* 4. accumulator = sum_p FDL[(idx+p) mod P] * H[p]  (:270-350) */
    memset(e->acc_re, 0, sizeof(float) * (size_t)H);
    memset(e->acc_im, 0, sizeof(float) * (size_t)H);
    {   /* p = 0: direct write  (:293-311) */
        const float *fr = e->fdl_re + (size_t)e->fdl_index * H, *fi = e->fdl_im + (size_t)e->fdl_index * H;
        e->acc_re[0] = fr[0] * e->hrir_re[0];            /* DC  (:304) */
        e->acc_im[0] = fi[0] * e->hrir_im[0];            /* Nyquist  (:305) */
        vdsp_zvmul(fr + 1, fi + 1, e->hrir_re + 1, e->hrir_im + 1, e->acc_re + 1, e->acc_im + 1, H - 1);
    }
    for (int p = 1; p < P; ++p) {                        /* (:315-350) */
        int idx = e->fdl_index + p;
        if (idx >= P) idx -= P;                          /* modulo partitionCount, not pow2 (:318-323) */
        const float *fr = e->fdl_re + (size_t)idx * H, *fi = e->fdl_im + (size_t)idx * H;
        const float *hr = e->hrir_re + (size_t)p * H, *hi = e->hrir_im + (size_t)p * H;
        e->acc_re[0] += fr[0] * hr[0];                   /* (:336) */
        e->acc_im[0] += fi[0] * hi[0];                   /* (:337) */
        vdsp_zvmul(fr + 1, fi + 1, hr + 1, hi + 1, e->tmp_re + 1, e->tmp_im + 1, H - 1);  /* (:346) */
        vdsp_zvadd(e->tmp_re + 1, e->tmp_im + 1, e->acc_re + 1, e->acc_im + 1, H - 1);    /* (:347) */
    }
    /* 5. inverse FFT (:353), 6. scale by 0.25/fftSize (:356-358) */
    vdsp_fft_zrip_inverse(e->setup, e->acc_re, e->acc_im);
    const float scale = 0.25f / (float)e->fft_size;
    for (int i = 0; i < H; ++i) { e->acc_re[i] *= scale; e->acc_im[i] *= scale; }
    /* 7. unpack, keep the second half  (:361-366) */
    vdsp_ztoc(e->acc_re, e->acc_im, e->input_buffer, H);
    memcpy(output, e->input_buffer + B, sizeof(float) * (size_t)B);
}

void orc_engine_process_accumulate(orc_engine *e, const float *input, float *accumulator) {
    orc_engine_process(e, input, e->temp_output);        /* :390 */
    for (int i = 0; i < e->block_size; ++i) accumulator[i] += e->temp_output[i];   /* vDSP_vadd :393 */
}

void orc_engine_reset(orc_engine *e) {                   /* :397-407 */
    memset(e->input_buffer, 0, sizeof(float) * (size_t)e->fft_size);
    memset(e->input_overlap, 0, sizeof(float) * (size_t)e->block_size);
    const size_t tot = (size_t)e->partition_count_pow2 * (size_t)e->fft_half;
    memset(e->fdl_re, 0, sizeof(float) * tot);
    memset(e->fdl_im, 0, sizeof(float) * tot);
    e->fdl_index = 0;
}

/* ------------------------------------------------------------------------------------------
 * RealtimeAudioProcessor
 * ---------------------------------------------------------------------------------------- */
struct orc_realtime {
    int block_size, max_frames, n_renderers, fifo_capacity;
    orc_engine **left, **right;          /* VirtualSpeakerRenderer.convolverLeftEar/RightEar */
    float *pending_left, *pending_right, *block_left, *block_right;
    float **left_temp, **right_temp;
    float *fifo_left, *fifo_right;
    int pending_count, fifo_read_index, fifo_count;
};

static void realtime_reset_storage(orc_realtime *p) {    /* :129-139 */
    memset(p->pending_left, 0, sizeof(float) * (size_t)p->block_size);
    memset(p->pending_right, 0, sizeof(float) * (size_t)p->block_size);
    memset(p->block_left, 0, sizeof(float) * (size_t)p->block_size);
    memset(p->block_right, 0, sizeof(float) * (size_t)p->block_size);
    memset(p->fifo_left, 0, sizeof(float) * (size_t)p->fifo_capacity);
    memset(p->fifo_right, 0, sizeof(float) * (size_t)p->fifo_capacity);
    p->pending_count = 0; p->fifo_read_index = 0; p->fifo_count = 0;
}

orc_realtime *orc_realtime_create(const float *left_irs, const float *right_irs, int n_renderers,
                                  int ir_len, int block_size, int max_frames) {
    if (n_renderers < 0 || ir_len <= 0 || block_size <= 0 || max_frames <= 0) return NULL;
    orc_realtime *p = (orc_realtime *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->block_size = block_size; p->max_frames = max_frames; p->n_renderers = n_renderers;
    p->fifo_capacity = max_frames + block_size;          /* :41 */
    p->left = (orc_engine **)calloc((size_t)(n_renderers > 0 ? n_renderers : 1), sizeof(void *));
    p->right = (orc_engine **)calloc((size_t)(n_renderers > 0 ? n_renderers : 1), sizeof(void *));
    p->left_temp = (float **)calloc((size_t)(n_renderers > 0 ? n_renderers : 1), sizeof(void *));
    p->right_temp = (float **)calloc((size_t)(n_renderers > 0 ? n_renderers : 1), sizeof(void *));
    p->pending_left = (float *)calloc((size_t)block_size, sizeof(float));
    p->pending_right = (float *)calloc((size_t)block_size, sizeof(float));
    p->block_left = (float *)calloc((size_t)block_size, sizeof(float));
    p->block_right = (float *)calloc((size_t)block_size, sizeof(float));
    p->fifo_left = (float *)calloc((size_t)p->fifo_capacity, sizeof(float));
    p->fifo_right = (float *)calloc((size_t)p->fifo_capacity, sizeof(float));
    for (int r = 0; r < n_renderers; ++r) {
        p->left[r] = orc_engine_create(left_irs + (size_t)r * ir_len, ir_len, block_size);
        p->right[r] = orc_engine_create(right_irs + (size_t)r * ir_len, ir_len, block_size);
        p->left_temp[r] = (float *)calloc((size_t)block_size, sizeof(float));
        p->right_temp[r] = (float *)calloc((size_t)block_size, sizeof(float));
        if (!p->left[r] || !p->right[r]) { orc_realtime_destroy(p); return NULL; }
    }
    realtime_reset_storage(p);
    return p;
}

void orc_realtime_destroy(orc_realtime *p) {
    if (!p) return;
    for (int r = 0; r < p->n_renderers; ++r) {
        if (p->left) orc_engine_destroy(p->left[r]);
        if (p->right) orc_engine_destroy(p->right[r]);
        if (p->left_temp) free(p->left_temp[r]);
        if (p->right_temp) free(p->right_temp[r]);
    }
    free(p->left); free(p->right); free(p->left_temp); free(p->right_temp);
    free(p->pending_left); free(p->pending_right); free(p->block_left); free(p->block_right);
    free(p->fifo_left); free(p->fifo_right);
    free(p);
}

static void realtime_process_pending_block(orc_realtime *p) {   /* :141-172 */
    const int B = p->block_size;
    memset(p->block_left, 0, sizeof(float) * (size_t)B);
    memset(p->block_right, 0, sizeof(float) * (size_t)B);
    const int count = p->n_renderers < 2 ? p->n_renderers : 2;  /* min(renderers.count, 2)  :145 */
    for (int r = 0; r < count; ++r) {
        const float *input = r == 0 ? p->pending_left : p->pending_right;   /* :147 */
        orc_engine_process(p->left[r], input, p->left_temp[r]);             /* :149 */
        orc_engine_process(p->right[r], input, p->right_temp[r]);           /* :150 */
        for (int i = 0; i < B; ++i) p->block_left[i] += p->left_temp[r][i];     /* vadd :152-157 */
        for (int i = 0; i < B; ++i) p->block_right[i] += p->right_temp[r][i];   /* vadd :158-163 */
    }
    for (int i = 0; i < B; ++i) {                                /* :166-171 */
        const int w = (p->fifo_read_index + p->fifo_count) % p->fifo_capacity;
        p->fifo_left[w] = p->block_left[i];
        p->fifo_right[w] = p->block_right[i];
        p->fifo_count += 1;
    }
}

int orc_realtime_process(orc_realtime *p, const float *in_l, const float *in_r, float *out_l,
                         float *out_r, int frame_count) {
    if (frame_count <= 0) return 0;                              /* guard :84 */
    if (frame_count > p->max_frames) return -1;                  /* precondition :85 */
    int off = 0;
    while (off < frame_count) {                                  /* :88-116 */
        int copy = p->block_size - p->pending_count;
        if (copy > frame_count - off) copy = frame_count - off;
        memcpy(p->pending_left + p->pending_count, in_l + off, sizeof(float) * (size_t)copy);
        memcpy(p->pending_right + p->pending_count, (in_r ? in_r : in_l) + off, sizeof(float) * (size_t)copy);
        p->pending_count += copy;
        off += copy;
        if (p->pending_count == p->block_size) {
            realtime_process_pending_block(p);
            p->pending_count = 0;
        }
    }
    for (int i = 0; i < frame_count; ++i) {                      /* drain :174-190 */
        if (p->fifo_count > 0) {
            out_l[i] = p->fifo_left[p->fifo_read_index];
            out_r[i] = p->fifo_right[p->fifo_read_index];
            p->fifo_read_index = (p->fifo_read_index + 1) % p->fifo_capacity;
            p->fifo_count -= 1;
        } else {
            out_l[i] = 0; out_r[i] = 0;
        }
    }
    return 0;
}

void orc_realtime_reset(orc_realtime *p) {                       /* :121-127 */
    for (int r = 0; r < p->n_renderers; ++r) {
        orc_engine_reset(p->left[r]);
        orc_engine_reset(p->right[r]);
    }
    realtime_reset_storage(p);
}

/* ------------------------------------------------------------------------------------------
 * N-speaker spatializer (generalised downmix; SURVEY.md §3.1 last paragraph)
 * ---------------------------------------------------------------------------------------- */
struct orc_spatializer {
    int n_channels, n_renderers, block_size;
    int *channel_of;                 /* renderer -> input channel */
    orc_engine **left, **right;
    float *chan, *tmp, *block_left, *block_right;
};

orc_spatializer *orc_spatializer_create(const float *tracks, int n_tracks, int taps, int n_channels,
                                        const int32_t *left_track, const int32_t *right_track,
                                        int block_size) {
    if (!tracks || n_tracks <= 0 || taps <= 0 || n_channels <= 0) return NULL;
    orc_spatializer *s = (orc_spatializer *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->n_channels = n_channels; s->block_size = block_size;
    s->channel_of = (int *)calloc((size_t)n_channels, sizeof(int));
    s->left = (orc_engine **)calloc((size_t)n_channels, sizeof(void *));
    s->right = (orc_engine **)calloc((size_t)n_channels, sizeof(void *));
    s->chan = (float *)calloc((size_t)block_size, sizeof(float));
    s->tmp = (float *)calloc((size_t)block_size, sizeof(float));
    s->block_left = (float *)calloc((size_t)block_size, sizeof(float));
    s->block_right = (float *)calloc((size_t)block_size, sizeof(float));
    for (int c = 0; c < n_channels; ++c) {
        const int l = left_track[c], r = right_track[c];
        if (l < 0 || r < 0) continue;                            /* unmapped: skipped, HRIRManager.swift:370-372 */
        if (l >= n_tracks || r >= n_tracks) {                    /* HRIRError.invalidChannelMapping :375-379 */
            orc_spatializer_destroy(s);
            return NULL;
        }
        const int k = s->n_renderers;
        s->left[k] = orc_engine_create(tracks + (size_t)l * taps, taps, block_size);    /* :406 */
        s->right[k] = orc_engine_create(tracks + (size_t)r * taps, taps, block_size);   /* :407 */
        if (!s->left[k] || !s->right[k]) { s->n_renderers = k + 1; orc_spatializer_destroy(s); return NULL; }
        s->channel_of[k] = c;
        s->n_renderers = k + 1;
    }
    if (s->n_renderers == 0) { orc_spatializer_destroy(s); return NULL; }   /* :420-422 */
    return s;
}

void orc_spatializer_destroy(orc_spatializer *s) {
    if (!s) return;
    for (int k = 0; k < s->n_renderers; ++k) {
        orc_engine_destroy(s->left[k]);
        orc_engine_destroy(s->right[k]);
    }
    free(s->channel_of); free(s->left); free(s->right);
    free(s->chan); free(s->tmp); free(s->block_left); free(s->block_right);
    free(s);
}

int orc_spatializer_process(orc_spatializer *s, const float *in, float *out, int64_t frames) {
    const int B = s->block_size, C = s->n_channels;
    if (frames % B != 0) return -1;
    for (int64_t f0 = 0; f0 < frames; f0 += B) {
        memset(s->block_left, 0, sizeof(float) * (size_t)B);
        memset(s->block_right, 0, sizeof(float) * (size_t)B);
        for (int k = 0; k < s->n_renderers; ++k) {               /* renderer order = channel order */
            const int c = s->channel_of[k];
            for (int i = 0; i < B; ++i) s->chan[i] = in[(size_t)(f0 + i) * C + c];
            orc_engine_process(s->left[k], s->chan, s->tmp);
            for (int i = 0; i < B; ++i) s->block_left[i] += s->tmp[i];
            orc_engine_process(s->right[k], s->chan, s->tmp);
            for (int i = 0; i < B; ++i) s->block_right[i] += s->tmp[i];
        }
        for (int i = 0; i < B; ++i) {
            out[(size_t)(f0 + i) * 2] = s->block_left[i];
            out[(size_t)(f0 + i) * 2 + 1] = s->block_right[i];
        }
    }
    return 0;
}

void orc_spatializer_reset(orc_spatializer *s) {
    for (int k = 0; k < s->n_renderers; ++k) {
        orc_engine_reset(s->left[k]);
        orc_engine_reset(s->right[k]);
    }
}

int orc_spatializer_batch(const float *tracks, int n_tracks, int taps, int n_channels,
                          const int32_t *left_track, const int32_t *right_track, int block_size,
                          const float *in, float *out, int n_streams, int64_t frames, int threads) {
    int rc = 0;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int s = 0; s < n_streams; ++s) {
        orc_spatializer *sp = orc_spatializer_create(tracks, n_tracks, taps, n_channels, left_track,
                                                     right_track, block_size);
        if (!sp) { rc = -1; continue; }
        if (orc_spatializer_process(sp, in + (size_t)s * (size_t)frames * (size_t)n_channels,
                                    out + (size_t)s * (size_t)frames * 2, frames) != 0)
            rc = -1;
        orc_spatializer_destroy(sp);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * float64 truth and synthetic input
 * ---------------------------------------------------------------------------------------- */
void orc_direct_conv_f64(const float *x, int64_t frames, int64_t x_stride, const float *h, int taps,
                         double *y, int accumulate) {
    for (int64_t n = 0; n < frames; ++n) {
        double acc = 0.0;
        const int64_t kmax = n < (int64_t)taps - 1 ? n : (int64_t)taps - 1;
        for (int64_t k = 0; k <= kmax; ++k) acc += (double)h[k] * (double)x[(n - k) * x_stride];
        if (accumulate) y[n] += acc; else y[n] = acc;
    }
}

static uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

float orc_synth_value(uint64_t seed, uint64_t stream, uint64_t index) {
    const uint64_t key = (seed + stream) * 0x9E3779B97F4A7C15ull + index;
    const uint32_t top = (uint32_t)(splitmix64(key) >> 40);        /* 24 bits */
    return (float)top * (1.0f / 16777216.0f) - 0.5f;
}

void orc_synth_fill(float *dst, int n_streams, int64_t frames, int n_channels, uint64_t seed) {
    const uint64_t per = (uint64_t)frames * (uint64_t)n_channels;
    for (int s = 0; s < n_streams; ++s)
        for (uint64_t i = 0; i < per; ++i) dst[(uint64_t)s * per + i] = orc_synth_value(seed, (uint64_t)s, i);
}

/* ------------------------------------------------------------------------------------------
 * Parametric EQ
 * ---------------------------------------------------------------------------------------- */
int orc_biquad_make(int type, double gain_db, double f, double q, double fs, orc_biquad *out) {
    if (!isfinite(fs) || !(fs > 0)) return 1;                                     /* :36-38 */
    if (!isfinite(gain_db) || !isfinite(f) || !isfinite(q)) return 4;             /* :39-41 */
    if (!(f > 0) || !(f < fs / 2)) return 2;                                      /* :42-44 */
    if (!(q > 0)) return 3;                                                       /* :45-47 */
    const double A = pow(10.0, gain_db / 40.0);                                   /* :49 */
    const double omega = 2.0 * M_PI * f / fs;
    const double sn = sin(omega), cs = cos(omega);
    const double alpha = sn / (2.0 * q);
    const double beta = 2.0 * sqrt(A) * alpha;
    double b0, b1, b2, a0, a1, a2;
    if (type == 0) {                                                              /* peaking :57-65 */
        b0 = 1 + alpha * A; b1 = -2 * cs; b2 = 1 - alpha * A;
        a0 = 1 + alpha / A; a1 = -2 * cs; a2 = 1 - alpha / A;
    } else if (type == 1) {                                                       /* lowShelf :66-74 */
        b0 = A * ((A + 1) - (A - 1) * cs + beta);
        b1 = 2 * A * ((A - 1) - (A + 1) * cs);
        b2 = A * ((A + 1) - (A - 1) * cs - beta);
        a0 = (A + 1) + (A - 1) * cs + beta;
        a1 = -2 * ((A - 1) + (A + 1) * cs);
        a2 = (A + 1) + (A - 1) * cs - beta;
    } else {                                                                      /* highShelf :75-83 */
        b0 = A * ((A + 1) + (A - 1) * cs + beta);
        b1 = -2 * A * ((A - 1) + (A + 1) * cs);
        b2 = A * ((A + 1) + (A - 1) * cs - beta);
        a0 = (A + 1) - (A - 1) * cs + beta;
        a1 = 2 * ((A - 1) - (A + 1) * cs);
        a2 = (A + 1) - (A - 1) * cs - beta;
    }
    if (!isfinite(a0) || a0 == 0) return 5;                                       /* :86-88 */
    out->b0 = b0 / a0; out->b1 = b1 / a0; out->b2 = b2 / a0; out->a1 = a1 / a0; out->a2 = a2 / a0;
    if (!isfinite(out->b0) || !isfinite(out->b1) || !isfinite(out->b2) || !isfinite(out->a1) || !isfinite(out->a2))
        return 5;                                                                 /* :97-104 */
    return 0;
}

struct orc_eq_state {
    double sample_rate, preamp_linear;
    int count;
    orc_biquad c[64];
    double lz1[64], lz2[64], rz1[64], rz2[64];
};

orc_eq_state *orc_eq_state_create(double fs, double preamp_db, const orc_biquad *c, int count) {
    if (count < 0 || count > 64) return NULL;                                     /* maximumFilterCount :17 */
    orc_eq_state *s = (orc_eq_state *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->sample_rate = fs; s->count = count;
    s->preamp_linear = pow(10.0, preamp_db / 20.0);                               /* :33 */
    for (int i = 0; i < count; ++i) s->c[i] = c[i];
    return s;
}
void orc_eq_state_destroy(orc_eq_state *s) { free(s); }
void orc_eq_state_reset(orc_eq_state *s) {                                        /* :49-56 */
    for (int i = 0; i < s->count; ++i) s->lz1[i] = s->lz2[i] = s->rz1[i] = s->rz2[i] = 0.0;
}
static inline double flush_subnormal(double v) { return fabs(v) < 1e-30 ? 0.0 : v; }   /* :95-97 */

void orc_eq_state_process(orc_eq_state *s, const float *in_l, const float *in_r, float *out_l, float *out_r, int n) {
    for (int f = 0; f < n; ++f) {                                                 /* :66-90 */
        double left = (double)in_l[f] * s->preamp_linear;
        double right = (double)(in_r ? in_r[f] : in_l[f]) * s->preamp_linear;
        for (int k = 0; k < s->count; ++k) {
            const orc_biquad c = s->c[k];
            const double lo = c.b0 * left + s->lz1[k];
            const double lz1 = c.b1 * left - c.a1 * lo + s->lz2[k];
            const double lz2 = c.b2 * left - c.a2 * lo;
            s->lz1[k] = flush_subnormal(lz1); s->lz2[k] = flush_subnormal(lz2);
            left = lo;
            const double ro = c.b0 * right + s->rz1[k];
            const double rz1 = c.b1 * right - c.a1 * ro + s->rz2[k];
            const double rz2 = c.b2 * right - c.a2 * ro;
            s->rz1[k] = flush_subnormal(rz1); s->rz2[k] = flush_subnormal(rz2);
            right = ro;
        }
        out_l[f] = (float)left;
        out_r[f] = (float)right;
    }
}
