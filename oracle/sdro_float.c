/* TEST INFRASTRUCTURE -- see sdro.h.  Float back-end of the oracle: NCO, Interpolator (polyphase
 * channel resampler), g_fft (John Green's radix-8 FFT as the reference uses it), fftfilt
 * (overlap-add FFT filter) and the FM discriminators.
 *
 * Build pinned to STRICT IEEE scalar arithmetic: -O2 -fno-fast-math -ffp-contract=off, no SSE2 path
 * (SURVEY.md finding 6).  Every expression keeps the reference's operand ORDER, because float
 * addition is not associative and the parity bar is <= 1 ulp.
 *
 * g_fft restatement (sdrbase/dsp/gfft.h; forward ffts1 :1189-1224, inverse iffts1 :2238-2275), for
 * sizes N = 2^M with (M-1) % 3 in {0, 1}: 16, 128, 1024 (fftfilt SSB), 8192 and 32, 256, 2048 (fftfilt
 * DSB), 16384.  Stage list: bit-reversed load fused with one radix-2 stage (bitrevR2 :185-317 /
 * scbitrevR2 :1231-1363, the latter scaling by 1/N); when (M-1) % 3 == 1 one more radix-2 stage with the
 * trivial twiddles 1 and -/+ i (bfR2 :531-635 / ibfR2 :1577-1681); then (M-1)/3 radix-8 passes (bfstages
 * :843-1158 / ibfstages :1889-2209).  The in-place index choreography of the reference has no effect on values;
 * only the arithmetic forms below do.  With multiplier m = (mr, mi):
 *     PLUS (a,b,m): r = (a.r + b.r*mr) - b.i*mi ;  i = (a.i + b.r*mi) + b.i*mr     ( = a + b*m )
 *     MINUS(a,b,m): r = (a.r - b.r*mr) + b.i*mi ;  i = (a.i - b.r*mi) - b.i*mr     ( = a - b*m )
 *     the partner is always formed as 2*a - result.
 * Forward multipliers are conj(w) resp. i*conj(w); inverse ones their conjugates (IEEE negation is
 * exact, so flipping the sign of mi reproduces the reference's explicit +/- forms bit for bit).
 */
#include "sdro.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float r, i; } cf;

/* ------------------------------------------------------------------ NCO (nco.cpp:30-64) */
#define NCO_N 4096
static float g_nco[NCO_N];
static int g_nco_ok = 0;
static void nco_init(void)
{
    if (g_nco_ok) return;
    for (int i = 0; i < NCO_N; i++) g_nco[i] = (float)cos((2.0 * 3.14159265358979323846 * i) / NCO_N);
    g_nco_ok = 1;
}
void sdro_nco_table(float* t) { nco_init(); memcpy(t, g_nco, sizeof g_nco); }
int32_t sdro_nco_inc(float freq, float rate) { return (int32_t)((freq * NCO_N) / rate); }   /* float math, trunc (:48-52) */

/* ------------------------------------------------------------------ Interpolator */
typedef struct {
    int phase_steps, ntaps;       /* ntaps = taps per phase */
    float* taps;                  /* [phase][ntaps] */
    cf* ring;                     /* ntaps entries; ptr = newest */
    int ptr;
} interp;

static void interp_create(interp* ip, int phase_steps, double sample_rate, double cutoff, double tpp)
{
    /* Interpolator::create -> createPolyphaseLowPass(taps, phaseSteps, 1.0, phaseSteps*sampleRate, cutoff, tpp)
     * (interpolator.cpp:21-56, 73-129) */
    const double M_PI_ = 3.14159265358979323846;
    double gain = 1.0;
    const double fs = phase_steps * sample_rate;
    int ntaps = (int)(tpp * phase_steps);
    if (ntaps % 2) ntaps++;
    ntaps *= phase_steps;
    float* taps = (float*)malloc(sizeof(float) * (size_t)ntaps);
    float* window = (float*)malloc(sizeof(float) * (size_t)ntaps);
    for (int n = 0; n < ntaps; n++) window[n] = (float)(0.54 - 0.46 * cos((2 * M_PI_ * n) / (ntaps - 1)));
    const int M = (ntaps - 1) / 2;
    const double fwT0 = 2 * M_PI_ * cutoff / fs;
    for (int n = -M; n <= M; n++) {
        if (n == 0) taps[n + M] = (float)(fwT0 / M_PI_ * window[n + M]);
        else taps[n + M] = (float)(sin(n * fwT0) / (n * M_PI_) * window[n + M]);
    }
    /* taps.resize() value-initialises: entries beyond 2M (ntaps even -> one) stay 0 */
    for (int n = 2 * M + 1; n < ntaps; n++) taps[n] = 0.0f;
    double mx = taps[0 + M];
    for (int n = 1; n <= M; n++) mx += 2.0 * taps[n + M];
    gain /= mx;
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);

    ip->phase_steps = phase_steps;
    ip->ntaps = ntaps / phase_steps;
    ip->taps = (float*)malloc(sizeof(float) * (size_t)ntaps);
    for (int ph = 0; ph < phase_steps; ph++)
        for (int i = 0; i < ip->ntaps; i++) ip->taps[ph * ip->ntaps + i] = taps[i * phase_steps + ph];
    for (int ph = 0; ph < phase_steps; ph++) {
        float sum = 0;
        for (int i = 0; i < ip->ntaps; i++) sum += ip->taps[ph * ip->ntaps + i];
        for (int i = 0; i < ip->ntaps; i++) ip->taps[ph * ip->ntaps + i] /= sum;
    }
    ip->ring = (cf*)calloc((size_t)ip->ntaps + 2, sizeof(cf));
    ip->ptr = 0;
    free(taps); free(window);
}

static void interp_free(interp* ip) { free(ip->taps); free(ip->ring); }

/* Interpolator::decimate (interpolator.h:23-36) with the scalar doInterpolate (:182-195) */
static int interp_decimate(interp* ip, float* distance, cf next, cf* result)
{
    ip->ptr--; if (ip->ptr < 0) ip->ptr = ip->ntaps - 1;          /* advanceFilter */
    ip->ring[ip->ptr] = next;
    *distance = (float)((double)*distance - 1.0);
    if (*distance >= 1.0) return 0;
    int phase = (int)floor(*distance * (float)ip->phase_steps);
    if (phase < 0) phase = 0;
    const float* c = ip->taps + phase * ip->ntaps;
    float ra = 0, ia = 0;
    int s = ip->ptr;
    for (int i = 0; i < ip->ntaps; i++) {
        ra += c[i] * ip->ring[s].r;
        ia += c[i] * ip->ring[s].i;
        s = (s + 1) % ip->ntaps;
    }
    result->r = ra; result->i = ia;
    return 1;
}

struct sdro_backend {
    int nco_inc, nco_phase;
    interp ip;
    float distance, step;
};

sdro_backend* sdro_backend_new(float nco_freq, float in_rate, float out_rate, int32_t phase_steps, float cutoff, float tpp)
{
    nco_init();
    sdro_backend* b = (sdro_backend*)calloc(1, sizeof *b);
    b->nco_inc = sdro_nco_inc(nco_freq, in_rate);
    b->nco_phase = 0;
    interp_create(&b->ip, phase_steps, in_rate, cutoff, tpp);
    b->distance = 0;
    b->step = in_rate / out_rate;                               /* (Real) inRate / (Real) outRate */
    return b;
}
void sdro_backend_free(sdro_backend* b) { if (b) { interp_free(&b->ip); free(b); } }
int32_t sdro_backend_ntaps(const sdro_backend* b) { return b->ip.ntaps; }
const float* sdro_backend_taps(const sdro_backend* b) { return b->ip.taps; }

int64_t sdro_backend_feed(sdro_backend* b, const int16_t* iq, int64_t n, float* out)
{
    int64_t n_out = 0;
    for (int64_t k = 0; k < n; k++) {
        /* NCO::nextIQ (nco.cpp:60-64): phase += inc, wrapped into [0, 4096) */
        b->nco_phase += b->nco_inc;
        while (b->nco_phase >= NCO_N) b->nco_phase -= NCO_N;
        while (b->nco_phase < 0) b->nco_phase += NCO_N;
        const float or_ = g_nco[b->nco_phase], oi = -g_nco[(b->nco_phase + NCO_N / 4) % NCO_N];
        /* Complex c(re, im); c *= osc;   std::complex<float> product: (ac - bd, ad + bc) */
        const float a = (float)iq[2 * k], bb = (float)iq[2 * k + 1];
        cf c; c.r = a * or_ - bb * oi; c.i = a * oi + bb * or_;
        cf ci;
        if (interp_decimate(&b->ip, &b->distance, c, &ci)) {
            out[2 * n_out] = ci.r; out[2 * n_out + 1] = ci.i; n_out++;
            b->distance += b->step;
        }
    }
    return n_out;
}

/* ------------------------------------------------------------------ g_fft */
typedef struct { int M, N; float* u; } gfft;

static int gfft_init(gfft* g, int n)
{
    int M = 0; while ((1 << M) < n) M++;
    if ((1 << M) != n || M < 4 || (M - 1) % 3 == 2) return -1;   /* (M-1)%3 == 2 would need the radix-4 stage bfR4 (not restated) */
    g->M = M; g->N = n;
    g->u = (float*)malloc(sizeof(float) * (size_t)(n / 4 + 1));
    /* fftCosInit (gfft.h:141-150): note the (float) casts of the index and of N */
    g->u[0] = 1.0f;
    for (int i = 1; i < n / 4; i++) g->u[i] = (float)cos((2.0 * 3.141592653589793238462643383279502884197 * (float)i) / (float)n);
    g->u[n / 4] = 0.0f;
    return 0;
}

static inline cf c_plus(cf a, cf b, float mr, float mi)  { cf t; t.r = (a.r + b.r * mr) - b.i * mi; t.i = (a.i + b.r * mi) + b.i * mr; return t; }
static inline cf c_minus(cf a, cf b, float mr, float mi) { cf t; t.r = (a.r - b.r * mr) + b.i * mi; t.i = (a.i - b.r * mi) - b.i * mr; return t; }
static inline cf c_two_minus(cf a, cf t)                 { cf f; f.r = a.r * 2.0f - t.r; f.i = a.i * 2.0f - t.i; return f; }

static unsigned bitrev(unsigned v, int bits) { unsigned r = 0; for (int b = 0; b < bits; b++) { r = (r << 1) | (v & 1u); v >>= 1; } return r; }

/* in-place transform of n complex floats; inverse != 0 -> scaled by 1/N in the first stage */
static void gfft_run(const gfft* g, cf* x, int inverse)
{
    const int N = g->N, M = g->M;
    cf* y = (cf*)malloc(sizeof(cf) * (size_t)N);
    /* stage 0: bit-reversed load + radix 2.  position 2j <- x[rev(2j)] + x[rev(2j) + N/2] */
    const float scale = (float)(1.0 / N);                    /* scbitrevR2's scale = 1.0/POW2(M) (gfft.h:2243) */
    for (int j = 0; j < N / 2; j++) {
        const unsigned r = bitrev((unsigned)(2 * j), M);
        const cf a = x[r], b = x[r + N / 2];
        cf s, d; s.r = a.r + b.r; s.i = a.i + b.i; d.r = a.r - b.r; d.i = a.i - b.i;
        if (inverse) { s.r = scale * s.r; s.i = scale * s.i; d.r = scale * d.r; d.i = scale * d.i; }
        y[2 * j] = s; y[2 * j + 1] = d;
    }
    int D0 = 2;
    if ((M - 1) % 3 == 1) {
        /* bfR2 / ibfR2: pairs (k, k+2) with twiddle 1 and (k+1, k+3) with -i (forward) or +i (inverse), k = 0 mod 4 */
        for (int k = 0; k < N; k += 4) {
            const cf a = y[k], b = y[k + 2], c = y[k + 1], d = y[k + 3];
            cf t;
            t.r = a.r + b.r; t.i = a.i + b.i; y[k] = t;
            t.r = a.r - b.r; t.i = a.i - b.i; y[k + 2] = t;
            if (!inverse) { t.r = c.r + d.i; t.i = c.i - d.r; y[k + 1] = t; t.r = c.r - d.i; t.i = c.i + d.r; y[k + 3] = t; }
            else          { t.r = c.r - d.i; t.i = c.i + d.r; y[k + 1] = t; t.r = c.r + d.i; t.i = c.i - d.r; y[k + 3] = t; }
        }
        D0 = 4;
    }
    /* radix-8 passes, D = D0, 8 D0, 64 D0, ... */
    const float sg = inverse ? 1.0f : -1.0f;                 /* forward multiplies by conj(w) */
    for (int D = D0; D < N; D *= 8) {
        const int uinc = N / 8 / D;                          /* table step of w2 per twiddle index */
        for (int u = 0; u < D; u++) {
            /* w0 = e^{j 4t}, w1 = e^{j 2t}, w2 = e^{j t}, w3 = e^{j (t + pi/4)}, t = 2 pi u / (8 D), read from the
             * quarter-wave cosine table; w0 crosses pi/2 at u = D/2 and is then mirrored with its cosine negated */
            const int i2 = u * uinc, i1 = 2 * i2;
            int i0 = 4 * i2; float w0r;
            if (u < D / 2) w0r = g->u[i0];
            else { i0 = N / 2 - i0; w0r = -g->u[i0]; }
            const float w0i = g->u[N / 4 - i0];
            const float w1r = g->u[i1], w1i = g->u[N / 4 - i1];
            const float w2r = g->u[i2], w2i = g->u[N / 4 - i2];
            const float w3r = g->u[i2 + N / 8], w3i = g->u[N / 4 - i2 - N / 8];
            for (int gidx = 0; gidx < N / 8 / D; gidx++) {
                cf* p = y + (size_t)gidx * 8 * D + u;
                cf f0 = p[0], f1 = p[D], f2 = p[2 * D], f3 = p[3 * D], f4 = p[4 * D], f5 = p[5 * D], f6 = p[6 * D], f7 = p[7 * D];
                cf t0, t1;
                /* layer 1 (w0) and layer 2 (w1), upper half */
                t0 = c_plus(f0, f1, w0r, sg * w0i);  f1 = c_two_minus(f0, t0);
                t1 = c_minus(f2, f3, w0r, sg * w0i); f2 = c_two_minus(f2, t1);
                f0 = c_plus(t0, f2, w1r, sg * w1i);  f2 = c_two_minus(t0, f0);
                f3 = c_plus(f1, t1, w1i, -sg * w1r); f1 = c_two_minus(f1, f3);
                /* lower half */
                t0 = c_plus(f4, f5, w0r, sg * w0i);  f5 = c_two_minus(f4, t0);
                t1 = c_minus(f6, f7, w0r, sg * w0i); f6 = c_two_minus(f6, t1);
                f4 = c_plus(t0, f6, w1r, sg * w1i);  f6 = c_two_minus(t0, f4);
                f7 = c_plus(f5, t1, w1i, -sg * w1r); f5 = c_two_minus(f5, f7);
                /* layer 3 (w2, w3) */
                t0 = c_minus(f0, f4, w2r, sg * w2i); f0 = c_two_minus(f0, t0);
                t1 = c_minus(f1, f5, w3r, sg * w3i); f1 = c_two_minus(f1, t1);
                p[4 * D] = t0; p[5 * D] = t1; p[0] = f0; p[D] = f1;
                { cf n4 = c_minus(f2, f6, w2i, -sg * w2r); f6 = c_two_minus(f2, n4); f4 = n4; }
                { cf n5 = c_minus(f3, f7, w3i, -sg * w3r); f7 = c_two_minus(f3, n5); f5 = n5; }
                p[2 * D] = f4; p[3 * D] = f5; p[6 * D] = f6; p[7 * D] = f7;
            }
        }
    }
    memcpy(x, y, sizeof(cf) * (size_t)N);
    free(y);
}

void sdro_gfft(float* iq, int32_t n, int32_t inverse)
{
    gfft g;
    if (gfft_init(&g, n) != 0) return;
    gfft_run(&g, (cf*)iq, inverse);
    free(g.u);
}

/* ------------------------------------------------------------------ fftfilt (fftfilt.cpp) */
struct sdro_fftfilt {
    int flen, flen2, inptr;
    gfft g;
    cf *filter, *filter_opp, *data, *ovl, *out;     /* filter_opp: create_asym_filter's opposite-band response */
};

static inline cf c_mul(cf a, cf b) { cf t; t.r = a.r * b.r - a.i * b.i; t.i = a.r * b.i + a.i * b.r; return t; }   /* std::complex<float> *= */

static float fsinc(float fc, int i, int len)
{
    /* fftfilt.h:57-62: double expression, returned as float */
    const int len2 = len / 2;
    return (i == len2) ? (float)(2.0 * fc)
                       : (float)(sin(2 * 3.14159265358979323846 * fc * (i - len2)) / (3.14159265358979323846 * (i - len2)));
}
static float blackman(int i, int len)
{
    return (float)(0.42 - 0.50 * cos(2.0 * 3.14159265358979323846 * i / len) + 0.08 * cos(4.0 * 3.14159265358979323846 * i / len));
}

sdro_fftfilt* sdro_fftfilt_new(float f1, float f2, int32_t len)
{
    sdro_fftfilt* f = (sdro_fftfilt*)calloc(1, sizeof *f);
    f->flen = len; f->flen2 = len >> 1;
    if (gfft_init(&f->g, len) != 0) { free(f); return 0; }
    f->filter = (cf*)calloc((size_t)len, sizeof(cf));
    f->data = (cf*)calloc((size_t)len, sizeof(cf));
    f->ovl = (cf*)calloc((size_t)f->flen2, sizeof(cf));
    f->out = (cf*)calloc((size_t)f->flen2, sizeof(cf));
    if (f1 < 0) {
        /* fftfilt(float f2, int len) -> create_dsb_filter (fftfilt.cpp:86-93, 149-170): low pass only */
        for (int i = 0; i < f->flen2; i++) { f->filter[i].r = fsinc(f2, i, f->flen2); f->filter[i].i = 0; }
    } else {
    /* create_filter (fftfilt.cpp:108-146) */
    const int lp = f2 != 0, hp = f1 != 0;
    for (int i = 0; i < f->flen2; i++) {
        f->filter[i].r = 0; f->filter[i].i = 0;
        if (lp) f->filter[i].r += fsinc(f2, i, f->flen2);
        if (hp) f->filter[i].r -= fsinc(f1, i, f->flen2);
    }
    if (hp && f2 < f1) f->filter[f->flen2 / 2].r += 1;
    }
    for (int i = 0; i < f->flen2; i++) { const float w = blackman(i, f->flen2); f->filter[i].r *= w; f->filter[i].i *= w; }
    gfft_run(&f->g, f->filter, 0);
    float scale = 0;
    for (int i = 0; i < f->flen2; i++) { const float mag = hypotf(f->filter[i].r, f->filter[i].i); if (mag > scale) scale = mag; }
    if (scale != 0) for (int i = 0; i < len; i++) { f->filter[i].r /= scale; f->filter[i].i /= scale; }
    return f;
}
/* low-pass design shared by create_dsb_filter and both halves of create_asym_filter (fftfilt.cpp:149-225):
 * windowed sinc in the first flen2 bins, forward FFT, normalise to max |H| over bins 0..flen2-1 */
static void design_lowpass(sdro_fftfilt* f, cf* dst, float fc)
{
    memset(dst, 0, sizeof(cf) * (size_t)f->flen);
    for (int i = 0; i < f->flen2; i++) { const float w = blackman(i, f->flen2); dst[i].r = fsinc(fc, i, f->flen2) * w; dst[i].i = 0.0f * w; }
    gfft_run(&f->g, dst, 0);
    float scale = 0;
    for (int i = 0; i < f->flen2; i++) { const float mag = hypotf(dst[i].r, dst[i].i); if (mag > scale) scale = mag; }
    if (scale != 0) for (int i = 0; i < f->flen; i++) { dst[i].r /= scale; dst[i].i /= scale; }
}

/* fftfilt(fin, len) followed by create_asym_filter(fopp, fin) (atvdemod.cpp:647): in-band and opposite-band low passes */
sdro_fftfilt* sdro_fftfilt_new_asym(float fopp, float fin, int32_t len)
{
    sdro_fftfilt* f = sdro_fftfilt_new(-1.0f, fin, len);
    if (!f) return 0;
    f->filter_opp = (cf*)calloc((size_t)len, sizeof(cf));
    design_lowpass(f, f->filter, fin);
    design_lowpass(f, f->filter_opp, fopp);
    return f;
}
const float* sdro_fftfilt_filter_opp(const sdro_fftfilt* f) { return (const float*)f->filter_opp; }

void sdro_fftfilt_free(sdro_fftfilt* f) { if (f) { free(f->g.u); free(f->filter); free(f->filter_opp); free(f->data); free(f->ovl); free(f->out); free(f); } }
const float* sdro_fftfilt_filter(const sdro_fftfilt* f) { return (const float*)f->filter; }

int64_t sdro_fftfilt_run(sdro_fftfilt* f, int32_t mode, const float* in, int64_t n, float* out)
{
    int64_t n_out = 0;
    const int h = f->flen2;
    for (int64_t k = 0; k < n; k++) {
        f->data[f->inptr].r = in[2 * k]; f->data[f->inptr].i = in[2 * k + 1];
        if (++f->inptr < h) continue;
        f->inptr = 0;
        gfft_run(&f->g, f->data, 0);
        if (mode == 0) {                                   /* runFilt (:261-283) */
            for (int i = 0; i < f->flen; i++) f->data[i] = c_mul(f->data[i], f->filter[i]);
        } else if (mode == 1 || mode == 2) {               /* runSSB (:285-325), getDC = true */
            f->data[0] = c_mul(f->data[0], f->filter[0]);
            if (mode == 1) for (int i = 1; i < h; i++) { f->data[i] = c_mul(f->data[i], f->filter[i]); f->data[h + i].r = 0; f->data[h + i].i = 0; }
            else           for (int i = 1; i < h; i++) { f->data[i].r = 0; f->data[i].i = 0; f->data[h + i] = c_mul(f->data[h + i], f->filter[h + i]); }
        } else if (mode == 4 || mode == 5) {               /* runAsym (:363-402): DC always kept, bin flen2 left as it is */
            f->data[0] = c_mul(f->data[0], f->filter[0]);
            if (mode == 4) for (int i = 1; i < h; i++) { f->data[i] = c_mul(f->data[i], f->filter[i]); f->data[h + i] = c_mul(f->data[h + i], f->filter_opp[h + i]); }
            else           for (int i = 1; i < h; i++) { f->data[i] = c_mul(f->data[i], f->filter_opp[i]); f->data[h + i] = c_mul(f->data[h + i], f->filter[h + i]); }
        } else {                                           /* runDSB (:327-361), getDC = true */
            for (int i = 0; i < h; i++) { f->data[i] = c_mul(f->data[i], f->filter[i]); f->data[h + i] = c_mul(f->data[h + i], f->filter[h + i]); }
        }
        gfft_run(&f->g, f->data, 1);
        for (int i = 0; i < h; i++) {
            out[2 * n_out] = f->ovl[i].r + f->data[i].r; out[2 * n_out + 1] = f->ovl[i].i + f->data[i].i; n_out++;
            f->ovl[i] = f->data[h + i];
        }
        memset(f->data, 0, sizeof(cf) * (size_t)f->flen);
    }
    return n_out;
}

/* ------------------------------------------------------------------ PhaseDiscriminators (phasediscri.h) */
static float atan2_approx2(float y, float x)
{
    /* phasediscri.h:172-197 */
    const float PI_F = 3.14159265f, PIBY2_F = 1.5707963f;
    if (x == 0.0f) { if (y > 0.0f) return PIBY2_F; if (y == 0.0f) return 0.0f; return -PIBY2_F; }
    float at;
    const float z = y / x;
    if (fabsf(z) < 1.0f) {
        at = z / (1.0f + 0.28f * z * z);
        if (x < 0.0f) { if (y < 0.0f) return at - PI_F; return at + PI_F; }
    } else {
        at = PIBY2_F - z / (z * z + 0.28f);
        if (y < 0.0f) return at - PI_F;
    }
    return at;
}

void sdro_discri(int32_t kind, float fm_scaling, const float* in, int64_t n, float* out)
{
    float prev_arg = 0; cf m1 = { 0, 0 };
    for (int64_t k = 0; k < n; k++) {
        const float I = in[2 * k], Q = in[2 * k + 1];
        if (kind == 0) {                                   /* phaseDiscriminatorDelta (:61-78) */
            const float cur = atan2_approx2(Q, I);
            float dev = (float)((cur - prev_arg) / 3.14159265358979323846);
            prev_arg = cur;
            if (dev < -1.0f) dev += 2.0f; else if (dev > 1.0f) dev -= 2.0f;
            out[k] = dev * fm_scaling;
        } else {                                           /* phaseDiscriminator (:50-55): conj(prev) * cur */
            cf d; d.r = m1.r * I - (-m1.i) * Q; d.i = m1.r * Q + (-m1.i) * I;
            m1.r = I; m1.i = Q;
            out[k] = (float)((atan2f(d.i, d.r) / 3.14159265358979323846) * fm_scaling);
        }
    }
}

/* ------------------------------------------------------------------ Lowpass<Real> / Bandpass<Real> (lowpass.h, bandpass.h)
 * Tap design restated with the reference's float/double mix; filter() restated from what the ring walk
 * actually sums (lowpass.h:55-99): with x[n] the newest sample, N taps, h = N/2,
 *     y[n] = (x[n] + x[n-1]) t[0] + sum_{i=1}^{h-1} (x[n-N+i] + x[n-1-i]) t[i] + x[n-N+h] t[h]
 * accumulated in that order in float (x[n-N] has just been overwritten by x[n], hence the odd first pair). */
struct sdro_fir { int N, h; float* t; float* hist; };

sdro_fir* sdro_fir_new(int32_t kind, int32_t ntaps, double rate, double f1, double f2)
{
    const double PI = 3.14159265358979323846;
    if (!(ntaps & 1)) ntaps++;
    sdro_fir* f = (sdro_fir*)calloc(1, sizeof *f);
    f->N = ntaps; f->h = ntaps / 2;
    const int nt = ntaps / 2 + 1;
    f->t = (float*)calloc((size_t)nt, sizeof(float));
    f->hist = (float*)calloc((size_t)ntaps, sizeof(float));
    const double mid = ((double)ntaps - 1.0) / 2.0;
    if (kind == 0) {                                           /* Lowpass::create (lowpass.h:16-52) */
        const double Wc = 2.0 * PI * f1 / rate;
        for (int i = 0; i < nt; i++)
            f->t[i] = (i == (ntaps - 1) / 2) ? (float)(Wc / PI) : (float)(sin(((double)i - mid) * Wc) / (((double)i - mid) * PI));
        for (int i = 0; i < nt; i++) f->t[i] = (float)(f->t[i] * (0.54 + 0.46 * cos((2.0 * PI * ((double)i - mid)) / (double)ntaps)));
    } else {                                                   /* Bandpass::create (bandpass.h:14-75) */
        const double Wcl = 2.0 * PI * f1 / rate, Wch = 2.0 * PI * f2 / rate;
        float* lp = (float*)calloc((size_t)nt, sizeof(float)); float* hp = (float*)calloc((size_t)nt, sizeof(float));
        for (int i = 0; i < nt; i++) {
            if (i == (ntaps - 1) / 2) { lp[i] = (float)(Wch / PI); hp[i] = (float)(-(Wcl / PI)); }
            else { lp[i] = (float)(sin(((double)i - mid) * Wch) / (((double)i - mid) * PI)); hp[i] = (float)(-sin(((double)i - mid) * Wcl) / (((double)i - mid) * PI)); }
        }
        hp[(ntaps - 1) / 2] += 1;
        for (int i = 0; i < nt; i++) {
            const double w = 0.54 + 0.46 * cos((2.0 * PI * ((double)i - mid)) / (double)ntaps);
            lp[i] = (float)(lp[i] * w); hp[i] = (float)(hp[i] * w);
            f->t[i] = -(lp[i] + hp[i]);
        }
        f->t[(ntaps - 1) / 2] += 1;
        free(lp); free(hp);
    }
    float sum = 0; int i;
    for (i = 0; i < nt - 1; i++) sum += f->t[i] * 2;
    sum += f->t[i];
    for (i = 0; i < nt; i++) f->t[i] /= sum;
    return f;
}
void sdro_fir_free(sdro_fir* f) { if (f) { free(f->t); free(f->hist); free(f); } }
int32_t sdro_fir_taps(const sdro_fir* f, float* out) { memcpy(out, f->t, sizeof(float) * (size_t)(f->h + 1)); return f->h + 1; }

void sdro_fir_run(sdro_fir* f, const float* in, int64_t n, float* out)
{
    const int N = f->N, h = f->h;
    float* x = (float*)malloc(sizeof(float) * (size_t)(n + N));    /* x[N + k] = in[k]; x[0..N) = history (oldest first) */
    memcpy(x, f->hist, sizeof(float) * (size_t)N);
    memcpy(x + N, in, sizeof(float) * (size_t)n);
    for (int64_t k = 0; k < n; k++) {
        const float* c = x + N + k;                            /* c[0] = x[n], c[-1] = x[n-1], ... */
        float acc = 0;
        acc += (c[0] + c[-1]) * f->t[0];
        for (int i = 1; i < h; i++) acc += (c[-N + i] + c[-1 - i]) * f->t[i];
        acc += c[-N + h] * f->t[h];
        out[k] = acc;
    }
    memcpy(f->hist, x + n, sizeof(float) * (size_t)N);
    free(x);
}
