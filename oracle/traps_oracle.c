/* traps_oracle.c -- see traps_oracle.h.  TEST INFRASTRUCTURE ONLY. */
#include "traps_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { SYS_3BT, SYS_1BT, SYS_1BT_DCT };

struct orc_traps {
    int system, nbanks, trap_bands, add_c0, hamming;
    int n_band;                 /* band nets: trap_bands for 1BT / 3BT, none for 1BT_DCT */
    orc_net *band;
    orc_net merger;
    int shift;                  /* merger_input_shift = merger inputs / trap_bands (traps.cpp:170) */
    float hamm[ORC_TRAP_LEN];   /* sSet(1) + sWindow_Hamming (traps.cpp:107-109, dspc.h:162-167) */
};

int orc_traps_create(orc_traps **out, const char *dir, const char *system, int nbanks, int add_c0, int hamming)
{
    orc_traps *t = calloc(1, sizeof *t);
    if (!t) return ORC_MEMORY;
    if (!strcmp(system, "3BT")) t->system = SYS_3BT;
    else if (!strcmp(system, "1BT")) t->system = SYS_1BT;
    else if (!strcmp(system, "1BT_DCT")) t->system = SYS_1BT_DCT;
    else { free(t); return ORC_CREATEERR; }
    t->nbanks = nbanks;
    t->add_c0 = add_c0;
    t->hamming = hamming;
    t->trap_bands = t->system == SYS_3BT ? nbanks - 2 : nbanks;          /* traps.cpp:95-97 */
    for (int i = 0; i < ORC_TRAP_LEN; i++)
        t->hamm[i] = 1.0f * (0.54f - 0.46f * cosf(2.0f * (float)M_PI * i / (ORC_TRAP_LEN - 1)));
    char fw[1024], fn[1024];
    if (t->system != SYS_1BT_DCT) {                                       /* traps.cpp:123-156 */
        t->n_band = t->trap_bands;
        t->band = calloc((size_t)t->n_band, sizeof(orc_net));
        for (int i = 0; i < t->n_band; i++) {
            snprintf(fw, sizeof fw, "%s/weights/band%d.weights", dir, i);
            snprintf(fn, sizeof fn, "%s/norms/band%d.norms", dir, i);
            int rc = orc_net_load(&t->band[i], fw, fn, 0);
            if (rc) { orc_traps_destroy(t); return rc; }
        }
    }
    snprintf(fw, sizeof fw, "%s/weights/merger.weights", dir);
    snprintf(fn, sizeof fn, "%s/norms/merger.norms", dir);
    int rc = orc_net_load(&t->merger, fw, fn, 0);
    if (rc) { orc_traps_destroy(t); return rc; }
    t->shift = t->merger.nInp / t->trap_bands;
    *out = t;
    return ORC_OK;
}

void orc_traps_destroy(orc_traps *t)
{
    if (!t) return;
    for (int i = 0; i < t->n_band; i++) orc_net_free(&t->band[i]);
    free(t->band);
    orc_net_free(&t->merger);
    free(t);
}

int orc_traps_num_outputs(const orc_traps *t) { return t->merger.nOut; }
int orc_traps_num_band_nets(const orc_traps *t) { return t->n_band; }

/* sDCT dspc.h:206-221 and CalcC0 dspc.h:223-233 over one 31-point trajectory */
static void dct31(const float *re, int add_c0, int shift, float *out)
{
    const int n = ORC_TRAP_LEN;
    const float NormC = sqrtf(2.0f / (float)n), PiByN = (float)M_PI / (float)n;
    int nOut = shift;
    if (add_c0) {
        float sum = 0.0f;
        for (int i = 0; i < n; i++) sum += re[i];
        sum *= NormC;
        *out++ = sum;
        nOut = shift - 1;
    }
    for (int k = 0; k < nOut; k++) {
        float acc = 0;
        const float v = PiByN * (float)(k + 1);
        for (int j = 0; j < n; j++) acc += re[j] * cosf(v * ((float)j + 0.5f));
        acc *= NormC;
        out[k] = acc;
    }
}

/* One frame: CalcInputFeaturesForBandNets (traps.cpp:220-283), ForwardPassBandNets (:347-358),
 * CalcInputFeaturesForMerger (:409-433: concat, sLn dspc.h:155-160, times -1), ForwardPassMerger (:465). */
static void one_frame(const orc_traps *t, const float *ctx /* [nbanks][31] */, float *g, float *post)
{
    float x[ORC_TRAP_LEN];
    const int n = ORC_TRAP_LEN;
    float *outp = g;
    for (int b = 0; b < t->trap_bands; b++) {
        for (int j = 0; j < n; j++) x[j] = t->hamming ? ctx[b * n + j] * t->hamm[j] : ctx[b * n + j];
        if (t->system == SYS_1BT_DCT) {
            dct31(x, t->add_c0, t->shift, outp);
            outp += t->shift;
        } else {
            const orc_net *net = &t->band[b];
            /* the reference copies trap_len values per band whatever the net's input size is (traps.cpp:257);
             * nets of these systems take 31 inputs */
            float in[ORC_TRAP_LEN];
            memcpy(in, x, sizeof in);
            orc_net_forward(net, in, outp, 1);
            outp += net->nOut;
        }
    }
    if (t->system != SYS_1BT_DCT) {
        const int K = t->merger.nInp;
        for (int i = 0; i < K; i++) {
            float v = g[i] > 0.0f ? logf(g[i]) : 0.0f;      /* sLn */
            g[i] = v * -1.0f;                               /* sMultiplication(.., -1) */
        }
    }
    orc_net_forward(&t->merger, g, post, 1);
}

void orc_traps_posteriors_batch(const orc_traps *t, const float *mel, const int *off, int n_utts,
                                float *post, float *merger_in)
{
    const int nb = t->nbanks, O = t->merger.nOut, K = t->merger.nInp;
    float *ctx = malloc(sizeof(float) * (size_t)nb * ORC_TRAP_LEN);
    float *g = malloc(sizeof(float) * (size_t)(K + 16));
    for (int u = 0; u < n_utts; u++) {
        const int a = off[u], e = off[u + 1];
        for (int r = a; r < e; r++) {
            for (int tap = 0; tap < ORC_TRAP_LEN; tap++) {
                int s = r - ORC_SHIFT + tap;
                if (s < a) s = a;
                if (s > e - 1) s = e - 1;
                for (int b = 0; b < nb; b++) ctx[b * ORC_TRAP_LEN + tap] = mel[(size_t)s * nb + b];
            }
            one_frame(t, ctx, g, post + (size_t)r * O);
            if (merger_in) memcpy(merger_in + (size_t)r * K, g, sizeof(float) * (size_t)K);
        }
    }
    free(ctx);
    free(g);
}
