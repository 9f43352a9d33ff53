/*
 * lcrc_oracle.c -- CPU restatement of PhnRec's LCRC posterior path.
 *
 * TEST INFRASTRUCTURE ONLY (see lcrc_oracle.h): the checker for the HIP path,
 * and the "port" CPU baseline of bench.py.  Never linked into the product.
 *
 * Written from the reference's behaviour, not from its text: the reference is
 * a streaming C++ class pair (Traps + NeuralNet) working on 5-row bunches; this
 * file states the same arithmetic as plain C functions over whole utterances,
 * plus a ring-buffer form used only to prove the two are the same function.
 * Arithmetic notes that matter for parity:
 *   - all sums are sequential f32 in the reference's loop order (compile with
 *     -ffp-contract=off so mul+add are never fused, as on the reference's x86
 *     -O2 build);
 *   - the hidden/output accumulators START from the bias (nn.cpp:857-870 then
 *     :784-787);
 *   - exp is the ICSI bit trick (fexp.h), sigmoid is evaluated in double
 *     because FEXP_EXP yields a double (fexp.h:33-38);
 *   - the low word of the FEXP double is 0 here; the reference leaves it
 *     uninitialised (fexp.h:23-31), worth <=2^-20 relative.
 */
#define _GNU_SOURCE
#include "lcrc_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_LN2
#define M_LN2 0.69314718055994530942
#endif
#ifndef M_PI
#define M_PI 3.1415926535897932384626433832795
#endif

/* ======================================================================== */
/* fexp.h                                                                    */
/* ======================================================================== */

/* FEXP_EXP (fexp.h:14-21): hi32 = (int)(2^20/ln2 * y) + (1072693248-60801),
 * lo32 = 0, reinterpret as double.  The (int) conversion is x86 cvttsd2si:
 * truncation, and INT_MIN ("integer indefinite") when out of range or NaN;
 * the add wraps.  Stated explicitly so no C undefined behaviour is involved. */
static inline double orc_fexp_d(float y)
{
    const double a = 1048576.0 / M_LN2;
    double t = a * (double)y;
    int32_t i;
    if (t > -2147483649.0 && t < 2147483648.0)
        i = (int32_t)t;
    else
        i = INT32_MIN;
    uint32_t hi = (uint32_t)i + (uint32_t)(1072693248 - 60801);
    uint64_t bits = (uint64_t)hi << 32;
    double d;
    memcpy(&d, &bits, sizeof d);
    return d;
}

float orc_fexp(float y) { return (float)orc_fexp_d(y); }

/* fexp_sigmoid (fexp.h:33-38): 1.0f/(1.0f+FEXP_EXP(-x)) -- the macro's value
 * is a double, so the add and the divide happen in double; one rounding to
 * float at the return. */
float orc_fexp_sigmoid(float x)
{
    return (float)(1.0 / (1.0 + orc_fexp_d(-x)));
}

/* fexp_softmax_v (fexp.h:49-78) */
void orc_fexp_softmax(int n, float *v)
{
    float mx = -FLT_MAX;
    for (int i = 0; i < n; i++)
        if (v[i] > mx) mx = v[i];
    float sum = 0.0f;
    for (int i = 0; i < n; i++) {
        v[i] = (float)orc_fexp_d(v[i] - mx);
        sum += v[i];
    }
    float scale = 1.0f / sum;
    for (int i = 0; i < n; i++)
        v[i] *= scale;
}

/* ======================================================================== */
/* nn.cpp -- storage                                                         */
/* ======================================================================== */

static int pad4(int n) { return (n + 3) & ~3; }   /* nn.cpp:633-651: x16 BYTES */

static int net_alloc(orc_net *net)
{
    net->nInp16 = pad4(net->nInp);
    net->nHid16 = pad4(net->nHid);
    net->nOut16 = pad4(net->nOut);
    net->W1 = calloc((size_t)net->nHid16 * net->nInp16, sizeof(float));
    net->W2 = calloc((size_t)net->nOut16 * net->nHid16, sizeof(float));
    net->b1 = calloc(net->nHid16, sizeof(float));
    net->b2 = calloc(net->nOut16, sizeof(float));
    net->mean = calloc(net->nInp16, sizeof(float));
    net->dev = calloc(net->nInp16, sizeof(float));
    if (!net->W1 || !net->W2 || !net->b1 || !net->b2 || !net->mean || !net->dev)
        return ORC_MEMORY;
    for (int i = 0; i < net->nInp16; i++) net->dev[i] = 1.0f;   /* nn.cpp:344-348 */
    return ORC_OK;
}

void orc_net_free(orc_net *net)
{
    free(net->W1); free(net->W2); free(net->b1); free(net->b2);
    free(net->mean); free(net->dev);
    memset(net, 0, sizeof *net);
}

/* .nbin (nn.cpp:464-531): int32 nlayers(=2), int32 nInp,nHid,nOut, then
 * W1[nHid16][nInp16], W2[nOut16][nHid16], b1[nHid16], b2[nOut16],
 * mean[nInp16], dev[nInp16], host byte order, no magic. */
int orc_net_load_nbin(orc_net *net, const char *path)
{
    memset(net, 0, sizeof *net);
    FILE *f = fopen(path, "rb");
    if (!f) return ORC_NOWEIGHTS;
    int32_t nl = 0, sz[3];
    if (fread(&nl, 4, 1, f) != 1 || nl != 2) { fclose(f); return ORC_BADWEIGHTS; }
    if (fread(sz, 4, 3, f) != 3) { fclose(f); return ORC_WRITEERR; }
    net->nInp = sz[0]; net->nHid = sz[1]; net->nOut = sz[2];
    if (net->nInp <= 0 || net->nHid <= 0 || net->nOut <= 0) { fclose(f); return ORC_BADWEIGHTS; }
    int rc = net_alloc(net);
    if (rc) { fclose(f); return rc; }
    size_t n1 = (size_t)net->nInp16 * net->nHid16, n2 = (size_t)net->nHid16 * net->nOut16;
    int ok = fread(net->W1, 4, n1, f) == n1 && fread(net->W2, 4, n2, f) == n2 &&
             fread(net->b1, 4, net->nHid16, f) == (size_t)net->nHid16 &&
             fread(net->b2, 4, net->nOut16, f) == (size_t)net->nOut16 &&
             fread(net->mean, 4, net->nInp16, f) == (size_t)net->nInp16 &&
             fread(net->dev, 4, net->nInp16, f) == (size_t)net->nInp16;
    fclose(f);
    if (!ok) { orc_net_free(net); return ORC_WRITEERR; }   /* sic: nn.cpp:500-525 */
    return ORC_OK;
}

/* nn.cpp:533-592 */
int orc_net_save_nbin(const orc_net *net, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return ORC_CREATEERR;
    int32_t hdr[4] = { 2, net->nInp, net->nHid, net->nOut };
    size_t n1 = (size_t)net->nInp16 * net->nHid16, n2 = (size_t)net->nHid16 * net->nOut16;
    int ok = fwrite(hdr, 4, 4, f) == 4 && fwrite(net->W1, 4, n1, f) == n1 &&
             fwrite(net->W2, 4, n2, f) == n2 &&
             fwrite(net->b1, 4, net->nHid16, f) == (size_t)net->nHid16 &&
             fwrite(net->b2, 4, net->nOut16, f) == (size_t)net->nOut16 &&
             fwrite(net->mean, 4, net->nInp16, f) == (size_t)net->nInp16 &&
             fwrite(net->dev, 4, net->nInp16, f) == (size_t)net->nInp16;
    fclose(f);
    return ok ? ORC_OK : ORC_WRITEERR;
}

/* ---- ASCII weights / norms (nn.cpp:116-462) ------------------------------
 * "weigvec <nHid*nInp>" values..., "weigvec <nOut*nHid>" values...,
 * "biasvec <nHid>" values..., "biasvec <nOut>" values...; whitespace-separated
 * tokens, floats read with "%e".  nInp = n1 / nHid (nn.cpp:194). */
static char *slurp(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *t = malloc((size_t)len + 1);
    if (t && fread(t, 1, (size_t)len, f) != (size_t)len) { free(t); t = NULL; }
    fclose(f);
    if (t) t[len] = 0;
    return t;
}

static char *next_tok(char **p)
{
    char *s = *p;
    while (*s && strchr(" \t\n\r", *s)) s++;
    if (!*s) return NULL;
    char *e = s;
    while (*e && !strchr(" \t\n\r", *e)) e++;
    if (*e) *e++ = 0;
    *p = e;
    return s;
}

static int expect_hdr(char **p, const char *kw, int *count)
{
    char *t = next_tok(p);
    if (!t || strncmp(t, kw, strlen(kw)) != 0) return 0;
    t = next_tok(p);
    return t && sscanf(t, "%d", count) == 1;
}

static int read_vals(char **p, float *dst, int n)
{
    for (int i = 0; i < n; i++) {
        char *t = next_tok(p);
        if (!t || sscanf(t, "%e", &dst[i]) != 1) return 0;
    }
    return 1;
}

int orc_net_load_ascii(orc_net *net, const char *weights, const char *norms)
{
    memset(net, 0, sizeof *net);
    char *txt = slurp(weights);
    if (!txt) return ORC_NOWEIGHTS;
    /* pass 1: sizes (GetInfo, nn.cpp:116-197) */
    char *copy = strdup(txt), *p = copy;
    int n1, n2, nb1, nb2, ok = 1;
    float dummy;
    ok = ok && expect_hdr(&p, "weigvec", &n1);
    for (int i = 0; ok && i < n1; i++) ok = next_tok(&p) != NULL;
    ok = ok && expect_hdr(&p, "weigvec", &n2);
    for (int i = 0; ok && i < n2; i++) ok = next_tok(&p) != NULL;
    ok = ok && expect_hdr(&p, "biasvec", &nb1);
    for (int i = 0; ok && i < nb1; i++) ok = next_tok(&p) != NULL;
    ok = ok && expect_hdr(&p, "biasvec", &nb2);
    for (int i = 0; ok && i < nb2; i++) ok = next_tok(&p) != NULL;
    free(copy);
    (void)dummy;
    if (!ok || nb1 <= 0 || nb2 <= 0) { free(txt); return ORC_BADWEIGHTS; }
    net->nOut = nb2; net->nHid = nb1; net->nInp = n1 / nb1;
    int rc = net_alloc(net);
    if (rc) { free(txt); return rc; }
    /* pass 2: values (ParseWeights, nn.cpp:199-338) */
    p = txt; ok = 1;
    int cnt;
    ok = ok && expect_hdr(&p, "weigvec", &cnt);
    for (int h = 0; ok && h < net->nHid; h++)
        ok = read_vals(&p, net->W1 + (size_t)h * net->nInp16, net->nInp);
    ok = ok && expect_hdr(&p, "weigvec", &cnt);
    for (int o = 0; ok && o < net->nOut; o++)
        ok = read_vals(&p, net->W2 + (size_t)o * net->nHid16, net->nHid);
    ok = ok && expect_hdr(&p, "biasvec", &cnt) && read_vals(&p, net->b1, net->nHid);
    ok = ok && expect_hdr(&p, "biasvec", &cnt) && read_vals(&p, net->b2, net->nOut);
    free(txt);
    if (!ok) { orc_net_free(net); return ORC_BADWEIGHTS; }
    /* norms (ParseNorms, nn.cpp:340-412): "vec <n>" means, "vec <n>" devs */
    if (norms) {
        txt = slurp(norms);
        if (!txt) { orc_net_free(net); return ORC_NONORMS; }
        p = txt;
        ok = expect_hdr(&p, "vec", &cnt) && read_vals(&p, net->mean, net->nInp) &&
             expect_hdr(&p, "vec", &cnt) && read_vals(&p, net->dev, net->nInp);
        free(txt);
        if (!ok) { orc_net_free(net); return ORC_BADWEIGHTS; }  /* sic: nn.cpp:452 */
    }
    return ORC_OK;
}

/* Load (nn.cpp:594-621): replace the weight file's suffix by ".nbin" and try
 * the binary; else ASCII, then cache the binary next to it. */
int orc_net_load(orc_net *net, const char *weights, const char *norms, int write_cache)
{
    char bin[1100];
    snprintf(bin, sizeof bin - 8, "%s", weights);
    char *dot = strrchr(bin, '.'), *slash = strrchr(bin, '/');
    if (dot && (!slash || dot > slash)) *dot = 0;
    strcat(bin, ".nbin");
    if (orc_net_load_nbin(net, bin) == ORC_OK) return ORC_OK;
    int rc = orc_net_load_ascii(net, weights, norms);
    if (rc == ORC_OK && write_cache) orc_net_save_nbin(net, bin);
    return rc;
}

/* ======================================================================== */
/* nn.cpp -- forward                                                         */
/* ======================================================================== */

/* One row through the net.  Follows ForwardPass1Bunch (nn.cpp:872-899) on the
 * padded buffers Forward builds (nn.cpp:915-919): the row is zero-padded to
 * nInp16, normalised over nInp16 (pads: (0-0)*1), both products run over the
 * padded widths with zero weights in the pads. */
static void net_row(const orc_net *net, const float *in, float *out, float *hid_out,
                    float *x, float *h, float *o)
{
    const int I = net->nInp16, H = net->nHid16, O = net->nOut16;
    memcpy(x, in, (size_t)net->nInp * sizeof(float));
    for (int j = net->nInp; j < I; j++) x[j] = 0.0f;
    for (int j = 0; j < I; j++) {            /* Normalize, nn.cpp:702-716 */
        x[j] -= net->mean[j];
        x[j] *= net->dev[j];
    }
    for (int j = 0; j < H; j++) {            /* PrepareBiases + naive product */
        float c = net->b1[j];                /* nn.cpp:857-870, 771-793       */
        const float *w = net->W1 + (size_t)j * I;
        for (int k = 0; k < I; k++) c += x[k] * w[k];
        h[j] = c;
    }
    for (int j = 0; j < net->nHid; j++) h[j] = orc_fexp_sigmoid(h[j]);   /* nn.cpp:796-820 */
    for (int j = net->nHid; j < H; j++) h[j] = 0.0f;
    for (int j = 0; j < O; j++) {
        float c = net->b2[j];
        const float *w = net->W2 + (size_t)j * H;
        for (int k = 0; k < H; k++) c += h[k] * w[k];
        o[j] = c;
    }
    orc_fexp_softmax(net->nOut, o);          /* nn.cpp:822-855 */
    memcpy(out, o, (size_t)net->nOut * sizeof(float));
    if (hid_out) memcpy(hid_out, h, (size_t)net->nHid * sizeof(float));
}

void orc_net_forward_probe(const orc_net *net, const float *in, float *out,
                           float *hidden, int n)
{
    float *x = malloc(sizeof(float) * (size_t)(net->nInp16 + net->nHid16 + net->nOut16));
    float *h = x + net->nInp16, *o = h + net->nHid16;
    for (int r = 0; r < n; r++)
        net_row(net, in + (size_t)r * net->nInp, out + (size_t)r * net->nOut,
                hidden ? hidden + (size_t)r * net->nHid : NULL, x, h, o);
    free(x);
}

void orc_net_forward(const orc_net *net, const float *in, float *out, int n)
{
    orc_net_forward_probe(net, in, out, NULL, n);
}

/* ======================================================================== */
/* traps.cpp -- LCRC                                                         */
/* ======================================================================== */

struct orc_lcrc {
    int nbanks;
    int ncoef;              /* nInp / nbanks = 11 with add_c0 */
    orc_net band[2];
    orc_net merger;
    float win[2][ORC_HALF];
    /* DCT basis as sDCT evaluates it (dspc.h:206-221): cos_tab[k][j] =
     * cosf(v_k * ((float)j + 0.5f)), v_k = (float)M_PI/16 * (float)(k+1) */
    float cos_tab[ORC_HALF][ORC_HALF];
    float normc;            /* sqrtf(2/16) */
    /* streaming state (Traps::be_mat, initialized, delay) */
    float *ring;            /* [nbanks][31] */
    int   ring_init;
    int   delay;
};

static int load_window(const char *path, float *w)   /* traps.cpp:549-570 */
{
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    int ok = 1;
    for (int i = 0; i < ORC_HALF && ok; i++) ok = fscanf(f, "%f", &w[i]) == 1;
    fclose(f);
    return ok;
}

static void lcrc_tables(orc_lcrc *c)
{
    float pibyn = (float)M_PI / (float)ORC_HALF;
    for (int k = 0; k < ORC_HALF; k++) {
        float v = pibyn * (float)(k + 1);
        for (int j = 0; j < ORC_HALF; j++)
            c->cos_tab[k][j] = cosf(v * ((float)j + 0.5f));
    }
    c->normc = sqrtf(2.0f / (float)ORC_HALF);
}

int orc_lcrc_create_from(orc_lcrc **out, int nbanks, orc_net band0, orc_net band1,
                         orc_net merger, const float *win0, const float *win1)
{
    orc_lcrc *c = calloc(1, sizeof *c);
    if (!c) return ORC_MEMORY;
    c->nbanks = nbanks;
    c->band[0] = band0; c->band[1] = band1; c->merger = merger;
    memcpy(c->win[0], win0, sizeof c->win[0]);
    memcpy(c->win[1], win1, sizeof c->win[1]);
    c->ncoef = band0.nInp / nbanks;
    c->ring = calloc((size_t)nbanks * ORC_TRAP_LEN + 1, sizeof(float));
    lcrc_tables(c);
    *out = c;
    return ORC_OK;
}

/* Traps::Init, LCRC branch (traps.cpp:88-171); file names from config.h:31-39 */
int orc_lcrc_create(orc_lcrc **out, const char *dir, int nbanks)
{
    orc_net n[3];
    float win[2][ORC_HALF];
    char fw[1024], fn[1024];
    memset(n, 0, sizeof n);
    for (int i = 0; i < 2; i++) {
        snprintf(fw, sizeof fw, "%s/weights/band%d.weights", dir, i);
        snprintf(fn, sizeof fn, "%s/norms/band%d.norms", dir, i);
        int rc = orc_net_load(&n[i], fw, fn, 0);
        if (rc) return rc;
        snprintf(fw, sizeof fw, "%s/windows/band%d.window", dir, i);
        if (!load_window(fw, win[i])) return ORC_BADWEIGHTS;
    }
    snprintf(fw, sizeof fw, "%s/weights/merger.weights", dir);
    snprintf(fn, sizeof fn, "%s/norms/merger.norms", dir);
    int rc = orc_net_load(&n[2], fw, fn, 0);
    if (rc) return rc;
    return orc_lcrc_create_from(out, nbanks, n[0], n[1], n[2], win[0], win[1]);
}

void orc_lcrc_destroy(orc_lcrc *c)
{
    if (!c) return;
    orc_net_free(&c->band[0]); orc_net_free(&c->band[1]); orc_net_free(&c->merger);
    free(c->ring);
    free(c);
}

int orc_lcrc_num_outputs(const orc_lcrc *c) { return c->merger.nOut; }
int orc_lcrc_num_inputs(const orc_lcrc *c)  { return c->band[0].nInp; }
int orc_lcrc_nbanks(const orc_lcrc *c)      { return c->nbanks; }
const orc_net *orc_lcrc_net(const orc_lcrc *c, int w) { return w == 2 ? &c->merger : &c->band[w]; }
const float *orc_lcrc_window(const orc_lcrc *c, int w) { return c->win[w]; }
int orc_lcrc_delay(const orc_lcrc *c) { return c->delay; }

/* One half-context -> [C0, DCT1..] (traps.cpp:306-342; CalcC0 dspc.h:223-233,
 * sDCT dspc.h:206-221, sMultVect dspc.h:107-112).  x = 16 taps of one band. */
static void half_project(const orc_lcrc *c, const float *x, const float *win, float *out)
{
    float xw[ORC_HALF];
    for (int j = 0; j < ORC_HALF; j++) xw[j] = x[j] * win[j];
    float sum = 0.0f;
    for (int j = 0; j < ORC_HALF; j++) sum += xw[j];
    sum *= c->normc;
    out[0] = sum;
    for (int k = 0; k < c->ncoef - 1; k++) {
        float acc = 0;
        for (int j = 0; j < ORC_HALF; j++) acc += xw[j] * c->cos_tab[k][j];
        acc *= c->normc;
        out[1 + k] = acc;
    }
}

/* CalcInputFeaturesForBandNets, LCRC branch (traps.cpp:285-343):
 * left = taps 0..15, right = taps 15..30 of each band's 31-point trajectory. */
void orc_lcrc_project(const orc_lcrc *c, const float *ctx, float *in0, float *in1)
{
    for (int b = 0; b < c->nbanks; b++) {
        const float *x = ctx + (size_t)b * ORC_TRAP_LEN;
        half_project(c, x, c->win[0], in0 + (size_t)b * c->ncoef);
        half_project(c, x + ORC_SHIFT, c->win[1], in1 + (size_t)b * c->ncoef);
    }
}

/* Everything after the context is assembled, for ONE frame:
 * ForwardPassBandNets (traps.cpp:347-404), CalcInputFeaturesForMerger LCRC
 * branch + sLn (traps.cpp:435-461, dspc.h:155-160), ForwardPassMerger (:465). */
typedef struct {
    float *ctx, *in0, *in1, *p0, *p1, *g, *x, *h, *o;
} frame_ws;

static frame_ws ws_alloc(const orc_lcrc *c)
{
    frame_ws w;
    int K = c->band[0].nInp, O = c->band[0].nOut;
    int mi = c->merger.nInp16 > c->band[0].nInp16 ? c->merger.nInp16 : c->band[0].nInp16;
    int mh = c->merger.nHid16 > c->band[0].nHid16 ? c->merger.nHid16 : c->band[0].nHid16;
    if (c->band[1].nHid16 > mh) mh = c->band[1].nHid16;
    int mo = c->merger.nOut16 > c->band[0].nOut16 ? c->merger.nOut16 : c->band[0].nOut16;
    size_t tot = (size_t)c->nbanks * ORC_TRAP_LEN + 2 * (size_t)K + 4 * (size_t)O + mi + mh + mo + 64;
    float *m = malloc(tot * sizeof(float));
    w.ctx = m; m += (size_t)c->nbanks * ORC_TRAP_LEN;
    w.in0 = m; m += K; w.in1 = m; m += K;
    w.p0 = m; m += O; w.p1 = m; m += O;
    w.g = m; m += 2 * O;
    w.x = m; m += mi; w.h = m; m += mh; w.o = m;
    return w;
}

static void frame_from_ctx(const orc_lcrc *c, frame_ws *w, float *post)
{
    const int O = c->band[0].nOut;
    orc_lcrc_project(c, w->ctx, w->in0, w->in1);
    net_row(&c->band[0], w->in0, w->p0, NULL, w->x, w->h, w->o);
    net_row(&c->band[1], w->in1, w->p1, NULL, w->x, w->h, w->o);
    memcpy(w->g, w->p0, (size_t)O * sizeof(float));
    memcpy(w->g + O, w->p1, (size_t)c->band[1].nOut * sizeof(float));
    for (int j = 0; j < c->merger.nInp; j++)
        w->g[j] = w->g[j] > 0.0f ? logf(w->g[j]) : 0.0f;
    net_row(&c->merger, w->g, post, NULL, w->x, w->h, w->o);
}

static void posteriors_range(const orc_lcrc *c, const float *mel, int n, int r0, int r1,
                             float *post, float *in0, float *in1, float *p0, float *p1, float *g)
{
    const int nb = c->nbanks, K = c->band[0].nInp, O = c->merger.nOut, Ob = c->band[0].nOut;
    frame_ws w = ws_alloc(c);
    for (int r = r0; r < r1; r++) {
        for (int t = 0; t < ORC_TRAP_LEN; t++) {
            int s = r - ORC_SHIFT + t;
            if (s < 0) s = 0;
            if (s > n - 1) s = n - 1;
            for (int b = 0; b < nb; b++)
                w.ctx[(size_t)b * ORC_TRAP_LEN + t] = mel[(size_t)s * nb + b];
        }
        frame_from_ctx(c, &w, post + (size_t)r * O);
        if (in0) memcpy(in0 + (size_t)r * K, w.in0, (size_t)K * sizeof(float));
        if (in1) memcpy(in1 + (size_t)r * K, w.in1, (size_t)K * sizeof(float));
        if (p0) memcpy(p0 + (size_t)r * Ob, w.p0, (size_t)Ob * sizeof(float));
        if (p1) memcpy(p1 + (size_t)r * Ob, w.p1, (size_t)Ob * sizeof(float));
        if (g) memcpy(g + (size_t)r * 2 * Ob, w.g, (size_t)2 * Ob * sizeof(float));
    }
    free(w.ctx);
}

void orc_lcrc_posteriors_probe(const orc_lcrc *c, const float *mel, int n, float *post,
                               float *in0, float *in1, float *p0, float *p1, float *g)
{
    posteriors_range(c, mel, n, 0, n, post, in0, in1, p0, p1, g);
}

void orc_lcrc_posteriors(const orc_lcrc *c, const float *mel, int n, float *post)
{
    posteriors_range(c, mel, n, 0, n, post, NULL, NULL, NULL, NULL, NULL);
}

void orc_lcrc_posteriors_batch(const orc_lcrc *c, const float *mel, const int *off,
                               int n_utts, float *post)
{
    for (int u = 0; u < n_utts; u++) {
        int a = off[u], b = off[u + 1];
        if (b > a)
            orc_lcrc_posteriors(c, mel + (size_t)a * c->nbanks, b - a,
                                post + (size_t)a * c->merger.nOut);
    }
}

typedef struct { const orc_lcrc *c; const float *mel; int n, r0, r1; float *post; } mt_job;

static void *mt_run(void *p)
{
    mt_job *j = p;
    posteriors_range(j->c, j->mel, j->n, j->r0, j->r1, j->post, NULL, NULL, NULL, NULL, NULL);
    return NULL;
}

void orc_lcrc_posteriors_mt(const orc_lcrc *c, const float *mel, int n, float *post, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    mt_job jobs[256];
    for (int t = 0; t < threads; t++) {
        jobs[t] = (mt_job){ c, mel, n, (int)((long long)n * t / threads),
                            (int)((long long)n * (t + 1) / threads), post };
        pthread_create(&th[t], NULL, mt_run, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}

/* ---- streaming form ------------------------------------------------------ */

void orc_lcrc_reset(orc_lcrc *c) { c->ring_init = 0; }   /* traps.cpp:174-177 */

/* AddVectorToBEMatrix (traps.cpp:180-219): first frame floods all 31 slots of
 * every band; afterwards the whole [nbanks*31] array slides left by one float
 * (so slot 30 of band b briefly holds slot 0 of band b+1) and slot 30 of each
 * band is overwritten with the new frame. */
static void ring_add(orc_lcrc *c, const float *frame)
{
    const int nb = c->nbanks, L = ORC_TRAP_LEN;
    if (!c->ring_init) {
        c->ring_init = 1;
        for (int b = 0; b < nb; b++)
            for (int j = 0; j < L; j++) c->ring[b * L + j] = frame[b];
        c->delay = 0;
    } else {
        memmove(c->ring, c->ring + 1, (size_t)(nb * L - 1) * sizeof(float));
        for (int b = 0; b < nb; b++) c->ring[b * L + L - 1] = frame[b];
        if (++c->delay > 9999) c->delay = 9999;
    }
}

/* CalcFeaturesBunched / CalcFeatures (traps.cpp:470-535).  `bunch` only
 * changes the grouping, never the values: every frame's output depends on the
 * ring contents at the time it was pushed. */
void orc_lcrc_push(orc_lcrc *c, const float *mel, int n, float *post, int needed, int bunch)
{
    (void)bunch;
    frame_ws w = ws_alloc(c);
    for (int i = 0; i < n; i++) {
        ring_add(c, mel + (size_t)i * c->nbanks);
        if (needed) {
            memcpy(w.ctx, c->ring, (size_t)c->nbanks * ORC_TRAP_LEN * sizeof(float));
            frame_from_ctx(c, &w, post + (size_t)i * c->merger.nOut);
        }
    }
    free(w.ctx);
}

/* srec.cpp:1035-1059: prime with 15 frames (short files: pad by repeating the
 * last), main block, flush by repeating the last frame min(15,n) times. */
void orc_lcrc_process_offline(orc_lcrc *c, const float *mel, int n, float *post, int bunch)
{
    const int nb = c->nbanks, O = c->merger.nOut, S = ORC_SHIFT;
    float *tmp = malloc((size_t)S * nb * sizeof(float));
    orc_lcrc_reset(c);
    if (n >= S) {
        orc_lcrc_push(c, mel, S, NULL, 0, bunch);
    } else {
        memcpy(tmp, mel, (size_t)n * nb * sizeof(float));
        for (int i = n; i < S; i++)
            memcpy(tmp + (size_t)i * nb, mel + (size_t)(n - 1) * nb, (size_t)nb * sizeof(float));
        orc_lcrc_push(c, tmp, S, NULL, 0, bunch);
    }
    if (n > S) orc_lcrc_push(c, mel + (size_t)S * nb, n - S, post, 1, bunch);
    int m = n > S ? S : n;
    for (int i = 0; i < m; i++)
        memcpy(tmp + (size_t)i * nb, mel + (size_t)(n - 1) * nb, (size_t)nb * sizeof(float));
    orc_lcrc_push(c, tmp, m, post + (size_t)(n - m) * O, 1, bunch);
    free(tmp);
}

/* SentenceBasedNormalization, mean part (srec.cpp:1500-1511): column sums are
 * sequential f32 (Mat::sumColumns, matrix.h:2101-2116); div(v) is
 * mul(1.0f/v) (matrix.h:245,532-537); sub(v) is add(-v) (matrix.h:194-199). */
void orc_sentence_mean_norm(float *mel, int n, int nbanks)
{
    for (int b = 0; b < nbanks; b++) {
        float sum = 0.0f;
        for (int r = 0; r < n; r++) sum += mel[(size_t)r * nbanks + b];
        float mean = sum * (1.0f / (float)n);
        for (int r = 0; r < n; r++) mel[(size_t)r * nbanks + b] += -mean;
    }
}
