/* phndec_oracle.c -- CPU restatement of PhnRec's phoneme-loop Viterbi decoder (decoder/type=phndec).
 * TEST INFRASTRUCTURE ONLY (see lcrc_oracle.h).  Parity status: PINNED -- fed with the logarithm of the
 * reference CLI's own posterior dumps it reproduces the reference's label files for the bundled test.raw
 * (tests/golden/<system>/test.{lop,rec}; tests/test_oracle.py).
 *
 * Follows phndec.cpp: Init :44-94, ProcessFrame :96-189, TimePruning :191-234, Done :236-303. */
#include <float.h>
#include <stdlib.h>

#include "lcrc_oracle.h"

typedef struct {
    int P, S, W, H, nframes, nlab;
    float wpen, prev_alpha;
    float *alpha, *halpha;
    int *prev, *len, *hphn, *hlen;
    int *start, *end, *phn;
    float *score;
} dec_t;

static void emit(dec_t *d, int start, int end, int phn, float score)
{
    d->start[d->nlab] = start; d->end[d->nlab] = end; d->phn[d->nlab] = phn; d->score[d->nlab] = score;
    d->nlab++;
}

static void time_pruning(dec_t *d)
{
    if (d->nframes < d->H) return;
    float best = -FLT_MAX;
    int blen = 1, bprev = 0;
    for (int i = 0; i < d->P; i++)
        for (int j = 1; j <= d->S; j++)
            if (d->alpha[i * d->W + j] > best) {
                best = d->alpha[i * d->W + j];
                blen = d->len[i * d->W + j];
                bprev = d->prev[i * d->W + j];
            }
    int offs = d->H - 1 - blen, phn = bprev;
    while (offs > 0) {
        const int l = d->hlen[offs];
        phn = d->hphn[offs];
        offs -= l;
    }
    if (offs == 0) {
        const int end = d->nframes - d->H + 1, start = end - d->hlen[0];
        const float like = d->halpha[0] - d->prev_alpha;
        d->prev_alpha = d->halpha[0];
        if (phn >= 0) emit(d, start, end, phn, like);
    }
}

/* Decodes one utterance of T rows of (log) posteriors [T][cols]; labels go to start/end/phn/score
 * (capacity T each); returns their number. */
int orc_phndec(const float *logpost, int T, int cols, int P, int S, int prune, float wpen,
               int *start, int *end, int *phn, float *score)
{
    const float lh = -0.69314718055994530941723212145818f;
    dec_t d;
    d.P = P; d.S = S; d.W = S + 1; d.H = prune + 1; d.nframes = 0; d.nlab = 0;
    d.wpen = wpen; d.prev_alpha = 0.0f;
    d.start = start; d.end = end; d.phn = phn; d.score = score;
    d.alpha = malloc(sizeof(float) * (size_t)P * d.W);
    d.prev = malloc(sizeof(int) * (size_t)P * d.W);
    d.len = malloc(sizeof(int) * (size_t)P * d.W);
    d.hphn = malloc(sizeof(int) * (size_t)d.H);
    d.hlen = malloc(sizeof(int) * (size_t)d.H);
    d.halpha = malloc(sizeof(float) * (size_t)d.H);
    for (int i = 0; i < P * d.W; i++) { d.alpha[i] = -FLT_MAX; d.prev[i] = -1; d.len[i] = 0; }
    for (int i = 0; i < d.H; i++) { d.hphn[i] = -1; d.hlen[i] = -1; d.halpha[i] = -1.0f; }
    for (int i = 0; i < P; i++) d.alpha[i * d.W] = wpen;
    for (int t = 0; t < T; t++) {
        const float *f = logpost + (size_t)t * cols;
        for (int i = 0; i < P; i++) {
            float *a = d.alpha + i * d.W;
            int *pv = d.prev + i * d.W, *ln = d.len + i * d.W;
            for (int j = S; j > 0; j--) {
                const float stay = a[j] + lh, enter = a[j - 1] + lh, obs = f[i * S + (j - 1)];
                if (stay > enter) { a[j] = stay + obs; ln[j] += 1; }
                else { a[j] = enter + obs; pv[j] = pv[j - 1]; ln[j] = ln[j - 1] + 1; }
            }
        }
        float best = -FLT_MAX;
        int bi = 0;
        for (int i = 0; i < P; i++)
            if (d.alpha[i * d.W + S] > best) { best = d.alpha[i * d.W + S]; bi = i; }
        for (int k = 1; k < d.H; k++) { d.hphn[k - 1] = d.hphn[k]; d.hlen[k - 1] = d.hlen[k]; d.halpha[k - 1] = d.halpha[k]; }
        d.hphn[d.H - 1] = d.prev[bi * d.W + S];
        d.hlen[d.H - 1] = d.len[bi * d.W + S];
        d.halpha[d.H - 1] = best;
        for (int i = 0; i < P; i++) { d.alpha[i * d.W] = best + wpen; d.prev[i * d.W] = bi; d.len[i * d.W] = 0; }
        d.nframes++;
        time_pruning(&d);
    }
    /* Done() */
    int offs = d.H - 1, e = d.nframes, p = d.prev[0], first_tail = d.nlab;
    while (offs > 0 && p != -1) {
        const int len = d.hlen[offs], s = e - len;
        const float a = d.halpha[offs];
        const int pp = d.hphn[offs];
        offs -= len;
        const float like = offs > 0 ? a - d.halpha[offs] : a - d.prev_alpha;
        emit(&d, s, e, p, like);
        e = s;
        p = pp;
    }
    for (int i = first_tail, j = d.nlab - 1; i < j; i++, j--) {
        int t; float x;
        t = start[i]; start[i] = start[j]; start[j] = t;
        t = end[i]; end[i] = end[j]; end[j] = t;
        t = phn[i]; phn[i] = phn[j]; phn[j] = t;
        x = score[i]; score[i] = score[j]; score[j] = x;
    }
    free(d.alpha); free(d.prev); free(d.len); free(d.hphn); free(d.hlen); free(d.halpha);
    return d.nlab;
}
