/* traps_oracle.c -- see traps_oracle.h.  TEST INFRASTRUCTURE ONLY. */
#include "traps_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { SYS_3BT, SYS_1BT, SYS_1BT_DCT, SYS_LCRC };

#define ORC_MAX_TRAP_LEN 255

struct orc_traps {
    int system, nbanks, trap_bands, add_c0, hamming;
    int L, half;                /* posteriors/length (Traps::SetTrapLen) and (L - 1) / 2 + 1 (traps.cpp:93,288) */
    int n_band;                 /* band nets: trap_bands for 1BT / 3BT, 2 for LCRC, none for 1BT_DCT */
    orc_net *band;
    orc_net merger;
    int shift;                  /* merger_input_shift = merger inputs / trap_bands (traps.cpp:170) */
    float hamm[ORC_MAX_TRAP_LEN];   /* sSet(1) + sWindow_Hamming (traps.cpp:107-109, dspc.h:162-167) */
    float win[2 * ORC_MAX_TRAP_LEN]; /* LCRC: be_win, the two half contexts' windows (traps.cpp:150-155) */
};

/* LoadWindow (traps.cpp:549-570): `n` numbers read with fscanf("%f") */
static int load_window(const char *path, int n, float *w)
{
    FILE *f = fopen(path, "r");
    if (!f) return ORC_CREATEERR;
    for (int i = 0; i < n; i++)
        if (fscanf(f, "%f", &w[i]) != 1) { fclose(f); return ORC_CREATEERR; }
    fclose(f);
    return ORC_OK;
}

int orc_traps_create(orc_traps **out, const char *dir, const char *system, int nbanks, int add_c0, int hamming)
{
    return orc_traps_create_geometry(out, dir, system, nbanks, add_c0, hamming, ORC_TRAP_LEN);
}

int orc_traps_create_geometry(orc_traps **out, const char *dir, const char *system, int nbanks, int add_c0,
                              int hamming, int trap_len)
{
    if (trap_len < 2 || trap_len > ORC_MAX_TRAP_LEN) return ORC_CREATEERR;
    orc_traps *t = calloc(1, sizeof *t);
    if (!t) return ORC_MEMORY;
    if (!strcmp(system, "3BT")) t->system = SYS_3BT;
    else if (!strcmp(system, "1BT")) t->system = SYS_1BT;
    else if (!strcmp(system, "1BT_DCT")) t->system = SYS_1BT_DCT;
    else if (!strcmp(system, "LCRC")) t->system = SYS_LCRC;
    else { free(t); return ORC_CREATEERR; }
    t->nbanks = nbanks;
    t->add_c0 = add_c0;
    t->hamming = hamming;
    t->L = trap_len;
    t->half = (trap_len - 1) / 2 + 1;
    t->trap_bands = t->system == SYS_3BT ? nbanks - 2 : nbanks;          /* traps.cpp:95-97 */
    for (int i = 0; i < trap_len; i++)
        t->hamm[i] = 1.0f * (0.54f - 0.46f * cosf(2.0f * (float)M_PI * i / (trap_len - 1)));
    char fw[1024], fn[1024];
    if (t->system != SYS_1BT_DCT) {                                       /* traps.cpp:123-156 */
        t->n_band = t->system == SYS_LCRC ? 2 : t->trap_bands;
        t->band = calloc((size_t)t->n_band, sizeof(orc_net));
        for (int i = 0; i < t->n_band; i++) {
            snprintf(fw, sizeof fw, "%s/weights/band%d.weights", dir, i);
            snprintf(fn, sizeof fn, "%s/norms/band%d.norms", dir, i);
            int rc = orc_net_load(&t->band[i], fw, fn, 0);
            if (rc) { orc_traps_destroy(t); return rc; }
            if (t->system == SYS_LCRC) {
                snprintf(fw, sizeof fw, "%s/windows/band%d.window", dir, i);
                rc = load_window(fw, t->half, t->win + i * t->half);
                if (rc) { orc_traps_destroy(t); return rc; }
            }
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

/* sDCT dspc.h:206-221 and CalcC0 dspc.h:223-233 over one n-point trajectory: `shift` values */
static void dct_n(int n, const float *re, int add_c0, int shift, float *out)
{
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
static void one_frame(const orc_traps *t, const float *ctx /* be_mat: [nbanks][L] */, float *g, float *post)
{
    float x[ORC_MAX_TRAP_LEN];
    const int n = t->L;
    float *outp = g;
    if (t->system == SYS_LCRC) {
        /* traps.cpp:285-343.  LC / RC are cut from be_mat with a stride of 2 half - 1 per band (:296-306) -- the band's
         * own row for odd L, a walk that drifts over the rows for even L; restated as written, on the flat matrix. */
        const int H = t->half, K = t->band[0].nInp, nc = K / t->nbanks;
        float *in0 = malloc(sizeof(float) * (size_t)(2 * K + 32)), *in1 = in0 + K + 16;
        const float *inp = ctx;
        for (int b = 0; b < t->nbanks; b++) {
            float lc[ORC_MAX_TRAP_LEN], rc[ORC_MAX_TRAP_LEN];
            for (int j = 0; j < H; j++) { lc[j] = inp[j]; rc[j] = inp[j + (H - 1)]; }
            inp += 2 * H - 1;
            for (int j = 0; j < H; j++) { lc[j] = lc[j] * t->win[j]; rc[j] = rc[j] * t->win[H + j]; }   /* sMultVect */
            dct_n(H, lc, t->add_c0, nc, in0 + (size_t)b * nc);
            dct_n(H, rc, t->add_c0, nc, in1 + (size_t)b * nc);
        }
        orc_net_forward(&t->band[0], in0, outp, 1);
        orc_net_forward(&t->band[1], in1, outp + t->band[0].nOut, 1);
        free(in0);
        const int Km = t->merger.nInp;
        for (int i = 0; i < Km; i++) g[i] = g[i] > 0.0f ? logf(g[i]) : 0.0f;      /* sLn only, traps.cpp:458 */
        orc_net_forward(&t->merger, g, post, 1);
        return;
    }
    for (int b = 0; b < t->trap_bands; b++) {
        for (int j = 0; j < n; j++) x[j] = t->hamming ? ctx[b * n + j] * t->hamm[j] : ctx[b * n + j];
        if (t->system == SYS_1BT_DCT) {
            dct_n(n, x, t->add_c0, t->shift, outp);
            outp += t->shift;
        } else {
            const orc_net *net = &t->band[b];
            /* the reference copies trap_len values per band whatever the net's input size is (traps.cpp:257);
             * nets of these systems take trap_len inputs */
            float in[ORC_MAX_TRAP_LEN];
            memcpy(in, x, sizeof(float) * (size_t)n);
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
    const int L = t->L, back = L - 1 - (L - 1) / 2;   /* the output frame's tap: GetTrapShift() pushes behind it */
    float *ctx = malloc(sizeof(float) * ((size_t)nb * L + 1));
    float *g = malloc(sizeof(float) * (size_t)(K + 16));
    for (int u = 0; u < n_utts; u++) {
        const int a = off[u], e = off[u + 1];
        for (int r = a; r < e; r++) {
            for (int tap = 0; tap < L; tap++) {
                int s = r - back + tap;
                if (s < a) s = a;
                if (s > e - 1) s = e - 1;
                for (int b = 0; b < nb; b++) ctx[b * L + tap] = mel[(size_t)s * nb + b];
            }
            one_frame(t, ctx, g, post + (size_t)r * O);
            if (merger_in) memcpy(merger_in + (size_t)r * K, g, sizeof(float) * (size_t)K);
        }
    }
    free(ctx);
    free(g);
}
