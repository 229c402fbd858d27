/* TEST INFRASTRUCTURE -- CPU oracle for the float half-band decimators:
 *   DecimatorsFI (float I/Q in, int16 Sample out)   sdrbase/dsp/decimatorsfi.{h,cpp}   (AirspyHF thread)
 *   DecimatorsFF (float in, float FSample out)      sdrbase/dsp/decimatorsff.{h,cpp}
 *   DecimatorsIF<qint16,InputBits> (int16 in, float out)  sdrbase/dsp/decimatorsif.h
 * all built on IntHalfbandFilterEOF<64> (sdrbase/dsp/inthalfbandfiltereof.h:65-72,141-188).
 *
 * Restated, not transcribed.  One stage (myDecimate + doFIR) on a stream x[n], output k from x[2k], x[2k+1]:
 *     acc = 0;  for i in 0..15: acc = acc + (x[2k+1-2i] + x[2k-61+2i]) * c[i];      (float, in this order, no FMA)
 *     y[k] = acc + x[2k-30] * 0.5f
 * (the ring indices a = tip, b = tail of inthalfbandfiltereof.h:153-171 resolve to these samples; the `m_ptr % 2`
 * branch taken by myDecimate is always the odd one).  c = (float) of the order-64 Remez decimals
 * (hbfiltertraits.cpp:173-190, the same 16 numbers SURVEY a1 lists).
 *
 * Chains (call patterns of decimatorsfi.cpp; FF and IF have the same structure):
 *   _cen, /2^L : L stages on the raw stream.
 *   _inf/_sup, L = 1 : no filter, two outputs per 4 input samples (:53-93).
 *   _inf/_sup, L = 2 : no filter, one output per 4 input samples ("4x downsample and rotate", :95-131).
 *   _inf/_sup, L >= 3: that 4-sample combination first, then L-2 stages (:133-371).
 *   The float sums keep the source's association: inf re ((b0-b3)+b7)-b4, im ((b1-b5)+b2)-b6;
 *   sup re ((b1-b2)-b5)+b6, im ((-b0-b3)+b4)+b7 for L <= 3 but ((b4+b7)-b0)-b3 for L >= 4 (:207-216 vs :157-166).
 * Conversions: FI output (int16)(v * 32768.0) (double product, truncation; decimate1: v * 32768.0f);
 *   IF input int16 -> float exact, combinations in int arithmetic, output v * scaleIn (1/128, 1/2048, 1/32768).
 * Whole groups only, tail dropped; state = each stage's last 62 inputs.
 *
 * Pinned by tests/test_oracle_vs_ref.py against the compiled reference classes (oracle/ref_shim_f.cpp) and by
 * tests/golden/fdecim_golden.npz generated from them. */
#include "sdro.h"
#include <stdlib.h>
#include <string.h>

static const double HB64_DEC[16] = {
    -0.0004653050334792540416659067936677729449, 0.0007120490624526883919470643391491648799,
    -0.0012303473710125558716887983479182366864, 0.0019716520179919017584369012041634050547,
    -0.0029947484165425580261710170049127555103, 0.0043703902150498061263128590780979720876,
    -0.0061858352927315653213558022116558277048, 0.0085554408639278121950777489246320328675,
    -0.0116397924445187355563247066925214312505, 0.0156852221106748394852115069397768820636,
    -0.0211070832238078286147153761476147337817, 0.0286850846890029896607554604770484729670,
    -0.0400956173930921908055147184768429724500, 0.0597215923200692666572564348825835622847,
    -0.1036982054813635201195864965484361164272, 0.3175014394028848885298543791577685624361,
};

#define FD_MEM 62               /* inputs a stage looks back */

struct sdro_fdecim {
    int log2, fcpos, in_kind, out_kind, bits;
    int n_stages;               /* half-band stages after the front end */
    float c[16];
    float hist[6][2][FD_MEM];   /* per stage: last 62 inputs, I and Q */
};

sdro_fdecim* sdro_fdecim_new(int log2, int fcpos, int in_kind, int out_kind, int input_bits)
{
    if (log2 < 0 || log2 > 6 || fcpos < 0 || fcpos > 2) return NULL;
    sdro_fdecim* d = (sdro_fdecim*)calloc(1, sizeof *d);
    if (!d) return NULL;
    d->log2 = log2; d->fcpos = fcpos; d->in_kind = in_kind; d->out_kind = out_kind; d->bits = input_bits;
    d->n_stages = fcpos == SDRO_FC_CEN ? log2 : (log2 >= 3 ? log2 - 2 : 0);
    for (int i = 0; i < 16; i++) d->c[i] = (float)HB64_DEC[i];
    return d;
}
void sdro_fdecim_free(sdro_fdecim* d) { free(d); }
void sdro_fdecim_reset(sdro_fdecim* d) { memset(d->hist, 0, sizeof d->hist); }
/* one DecimatorsFI / FF / IF object runs every decimateK_x on the same six filters (m_decimator2 .. m_decimator64): cascade
 * stage s is member s whatever the variant, so another variant continues on what each filter saw last */
void sdro_fdecim_switch(sdro_fdecim* d, int log2, int fcpos)
{
    d->log2 = log2; d->fcpos = fcpos;
    d->n_stages = fcpos == SDRO_FC_CEN ? log2 : (log2 >= 3 ? log2 - 2 : 0);
}

int32_t sdro_fdecim_group(int log2, int fcpos)
{
    if (log2 == 0) return 2;
    if (log2 == 1) return fcpos == SDRO_FC_CEN ? 4 : 8;
    return 2 << log2;
}

/* one half-band stage: x = 62 history samples followed by n (even) new inputs; n/2 outputs to y; history updated */
static int hb_stage(const float* c, float* hist, const float* x, int n, float* y)
{
    const int n_out = n / 2;
    const float* s = x + FD_MEM;                /* s[j] = input j of this call; s[-1..-62] = history */
    for (int k = 0; k < n_out; k++) {
        float acc = 0.0f;
        for (int i = 0; i < 16; i++) acc = acc + (s[2 * k + 1 - 2 * i] + s[2 * k - 61 + 2 * i]) * c[i];
        y[k] = acc + s[2 * k - 30] * 0.5f;
    }
    memcpy(hist, s + n - FD_MEM, FD_MEM * sizeof(float));
    return n_out;
}

static float in_at(const sdro_fdecim* d, const void* in, long i)
{
    return d->in_kind == 0 ? ((const float*)in)[i] : (float)((const int16_t*)in)[i];
}

int32_t sdro_fdecim_process(sdro_fdecim* d, const void* in, int32_t n_elems, void* out)
{
    const int grp = sdro_fdecim_group(d->log2, d->fcpos);
    const long n_groups = n_elems / grp;
    if (n_groups <= 0) return 0;
    const int L = d->log2, fc = d->fcpos;
    /* ---- front end: the stream that enters the first half-band stage (or leaves, when there is none) */
    long n_pre;
    if (fc == SDRO_FC_CEN || L == 0) n_pre = n_groups * (grp / 2);
    else if (L == 1) n_pre = n_groups * 2;
    else n_pre = n_groups * (grp / 8);
    float* bufI = (float*)malloc((size_t)(FD_MEM + n_pre) * sizeof(float));
    float* bufQ = (float*)malloc((size_t)(FD_MEM + n_pre) * sizeof(float));
    float* pI = bufI + FD_MEM, *pQ = bufQ + FD_MEM;
    if (fc == SDRO_FC_CEN || L == 0) {
        for (long p = 0; p < n_pre; p++) { pI[p] = in_at(d, in, 2 * p); pQ[p] = in_at(d, in, 2 * p + 1); }
    } else if (d->in_kind == 1) {               /* integer sums are exact: association does not matter */
        const int16_t* b = (const int16_t*)in;
        for (long g = 0; g < n_pre / (L == 1 ? 2 : 1); g++) {
            const int16_t* q = b + 8 * g;
            if (L == 1) {
                if (fc == SDRO_FC_INF) { pI[2*g] = (float)(q[0] - q[3]); pQ[2*g] = (float)(q[1] + q[2]); pI[2*g+1] = (float)(q[7] - q[4]); pQ[2*g+1] = (float)(-q[5] - q[6]); }
                else                   { pI[2*g] = (float)(q[1] - q[2]); pQ[2*g] = (float)(-q[0] - q[3]); pI[2*g+1] = (float)(q[6] - q[5]); pQ[2*g+1] = (float)(q[4] + q[7]); }
            } else if (fc == SDRO_FC_INF) { pI[g] = (float)(q[0] - q[3] + q[7] - q[4]); pQ[g] = (float)(q[1] - q[5] + q[2] - q[6]); }
            else                          { pI[g] = (float)(q[1] - q[2] - q[5] + q[6]); pQ[g] = (float)(-q[0] - q[3] + q[4] + q[7]); }
        }
    } else {
        const float* b = (const float*)in;
        for (long g = 0; g < n_pre / (L == 1 ? 2 : 1); g++) {
            const float* q = b + 8 * g;
            if (L == 1) {
                if (fc == SDRO_FC_INF) { pI[2*g] = q[0] - q[3]; pQ[2*g] = q[1] + q[2]; pI[2*g+1] = q[7] - q[4]; pQ[2*g+1] = -q[5] - q[6]; }
                else                   { pI[2*g] = q[1] - q[2]; pQ[2*g] = -q[0] - q[3]; pI[2*g+1] = q[6] - q[5]; pQ[2*g+1] = q[4] + q[7]; }
            } else if (fc == SDRO_FC_INF) { pI[g] = ((q[0] - q[3]) + q[7]) - q[4]; pQ[g] = ((q[1] - q[5]) + q[2]) - q[6]; }
            else {
                pI[g] = ((q[1] - q[2]) - q[5]) + q[6];
                pQ[g] = L <= 3 ? ((-q[0] - q[3]) + q[4]) + q[7] : ((q[4] + q[7]) - q[0]) - q[3];
            }
        }
    }
    /* ---- half-band stages */
    long n = n_pre;
    float* tI = (float*)malloc((size_t)(FD_MEM + n_pre / 2 + 1) * sizeof(float));
    float* tQ = (float*)malloc((size_t)(FD_MEM + n_pre / 2 + 1) * sizeof(float));
    for (int s = 0; s < d->n_stages; s++) {
        memcpy(bufI, d->hist[s][0], FD_MEM * sizeof(float));
        memcpy(bufQ, d->hist[s][1], FD_MEM * sizeof(float));
        hb_stage(d->c, d->hist[s][0], bufI, (int)n, tI + FD_MEM);
        n = hb_stage(d->c, d->hist[s][1], bufQ, (int)n, tQ + FD_MEM);
        float* w;
        w = bufI; bufI = tI; tI = w; w = bufQ; bufQ = tQ; tQ = w;
        pI = bufI + FD_MEM; pQ = bufQ + FD_MEM;
    }
    /* ---- output conversion */
    for (long k = 0; k < n; k++) {
        float vI = pI[k], vQ = pQ[k];
        if (d->in_kind == 1) {                  /* DecimatorsIF: scaleIn, applied once (to the sum or to the chain's output) */
            const float sc = d->bits == 8 ? (float)(1.0 / 128.0) : d->bits == 12 ? (float)(1.0 / 2048.0) : d->bits == 16 ? (float)(1.0 / 32768.0) : 1.0f;
            vI = vI * sc; vQ = vQ * sc;
        }
        if (d->out_kind == 0) {                 /* DecimatorsFI: setReal(v * SDR_RX_SCALED); decimate1 multiplies in float */
            int16_t* o = (int16_t*)out;
            if (L == 0) { o[2*k] = (int16_t)(int32_t)(vI * 32768.0f); o[2*k+1] = (int16_t)(int32_t)(vQ * 32768.0f); }
            else        { o[2*k] = (int16_t)(int32_t)(vI * 32768.0);  o[2*k+1] = (int16_t)(int32_t)(vQ * 32768.0); }
        } else { float* o = (float*)out; o[2*k] = vI; o[2*k+1] = vQ; }
    }
    free(bufI); free(bufQ); free(tI); free(tQ);
    return (int32_t)n;
}
