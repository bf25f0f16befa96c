/*
 * ge_oracle_sim.c -- CPU restatement of the literal-similarity edges of the reference
 * (SURVEY.md 8f rank 4): Rdf2GrphConverter's compare loop, CompareJob, and the metrics under
 * J/util/similarity/.  TEST INFRASTRUCTURE ONLY (see ge_oracle.h).
 *
 * PARITY UNPINNED, twice over for three of the metrics: JaroWinkler, NormalizedLevenshtein and the n-gram profile
 * (ShingleBased) live in info.debatty:java-string-similarity, which the reference pulls in as version "RELEASE"
 * (pom.xml:48-52, i.e. whatever was newest at build time) and which is not under /root/reference.  They are restated
 * from the library's published algorithm (v1.x/2.0 sources as remembered; the two JaroWinkler values of its README
 * are known answers in tests/test_oracle_kat.py after narrowing to float, which is what the reference stores).
 * Token*, Numeric and Date* are the reference's own code and are followed line by line.
 *
 * Strings are sequences of UTF-16 code units, as java.lang.String.
 */
#include "ge_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef const uint16_t *ustr;

static int u_equal(ustr a, int32_t na, ustr b, int32_t nb) {
    return na == nb && (na == 0 || memcmp(a, b, sizeof(uint16_t) * (size_t)na) == 0);
}
static int32_t u_index_of(ustr s, int32_t n, uint16_t c) {
    for (int32_t i = 0; i < n; ++i) if (s[i] == c) return i;
    return -1;
}

/* ---- info.debatty.java.stringsimilarity.JaroWinkler (threshold 0.7, JW_COEF 0.1, THREE = 3) ---- */
double geo_sim_jarowinkler(ustr s1, int32_t n1, ustr s2, int32_t n2) {
    if (u_equal(s1, n1, s2, n2)) return 1.0;
    ustr mx = s1, mn = s2; int32_t nmx = n1, nmn = n2;
    if (!(n1 > n2)) { mx = s2; nmx = n2; mn = s1; nmn = n1; }
    int32_t range = nmx / 2 - 1; if (range < 0) range = 0;
    int32_t *match_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nmn + 1));
    uint8_t *flag = (uint8_t *)calloc((size_t)nmx + 1, 1);
    int32_t matches = 0;
    for (int32_t mi = 0; mi < nmn; ++mi) {
        match_idx[mi] = -1;
        int32_t lo = mi - range; if (lo < 0) lo = 0;
        int32_t hi = mi + range + 1; if (hi > nmx) hi = nmx;
        for (int32_t xi = lo; xi < hi; ++xi)
            if (!flag[xi] && mn[mi] == mx[xi]) { match_idx[mi] = xi; flag[xi] = 1; ++matches; break; }
    }
    uint16_t *ms1 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(matches + 1));
    uint16_t *ms2 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(matches + 1));
    int32_t si = 0;
    for (int32_t i = 0; i < nmn; ++i) if (match_idx[i] != -1) ms1[si++] = mn[i];
    si = 0;
    for (int32_t i = 0; i < nmx; ++i) if (flag[i]) ms2[si++] = mx[i];
    int32_t transpositions = 0;
    for (int32_t i = 0; i < matches; ++i) if (ms1[i] != ms2[i]) ++transpositions;
    int32_t prefix = 0;
    for (int32_t i = 0; i < nmn; ++i) { if (s1[i] == s2[i]) ++prefix; else break; }
    free(match_idx); free(flag); free(ms1); free(ms2);
    const float m = (float)matches;
    if (m == 0) return 0.0;
    /* float arithmetic throughout: m is a float and THREE an int */
    const float jf = ((m / (float)n1 + m / (float)n2) + (m - (float)(transpositions / 2)) / m) / (float)3;
    const double j = (double)jf;
    double jw = j;
    if (j > 0.7) {
        const double inv = 1.0 / (double)nmx;
        jw = j + (0.1 < inv ? 0.1 : inv) * (double)prefix * (1 - j);
    }
    return jw;
}

/* ---- Levenshtein.distance + NormalizedLevenshtein.similarity ---- */
int32_t geo_sim_levenshtein_distance(ustr s1, int32_t n1, ustr s2, int32_t n2) {
    if (u_equal(s1, n1, s2, n2)) return 0;
    if (n1 == 0) return n2;
    if (n2 == 0) return n1;
    int32_t *v0 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    int32_t *v1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    for (int32_t i = 0; i <= n2; ++i) v0[i] = i;
    for (int32_t i = 0; i < n1; ++i) {
        v1[0] = i + 1;
        for (int32_t j = 0; j < n2; ++j) {
            const int32_t cost = s1[i] == s2[j] ? 0 : 1;
            int32_t best = v1[j] + 1;
            if (v0[j + 1] + 1 < best) best = v0[j + 1] + 1;
            if (v0[j] + cost < best) best = v0[j] + cost;
            v1[j + 1] = best;
        }
        int32_t *t = v0; v0 = v1; v1 = t;
    }
    const int32_t d = v0[n2];
    free(v0); free(v1);
    return d;
}
static double sim_levenshtein(ustr s1, int32_t n1, ustr s2, int32_t n2) {
    if (u_equal(s1, n1, s2, n2)) return 1.0 - 0.0;
    const int32_t m_len = n1 > n2 ? n1 : n2;
    if (m_len == 0) return 1.0;
    return 1.0 - (double)geo_sim_levenshtein_distance(s1, n1, s2, n2) / (double)m_len;
}

/* ---- profiles: a multiset of grams, kept as sorted (gram, count) ---- */
typedef struct { uint16_t *text; int32_t len; int32_t count; } gram;
typedef struct { gram *g; int32_t n, cap; } profile;

static int gram_cmp(const gram *a, const gram *b) {
    const int32_t n = a->len < b->len ? a->len : b->len;
    for (int32_t i = 0; i < n; ++i) if (a->text[i] != b->text[i]) return a->text[i] < b->text[i] ? -1 : 1;
    return a->len < b->len ? -1 : a->len > b->len;
}
static void profile_add(profile *p, const uint16_t *t, int32_t len) {   /* HashMap.merge(token, 1, Integer::sum) */
    gram key = {(uint16_t *)t, len, 1};
    for (int32_t i = 0; i < p->n; ++i) if (gram_cmp(&p->g[i], &key) == 0) { ++p->g[i].count; return; }
    if (p->n == p->cap) { p->cap = p->cap ? 2 * p->cap : 8; p->g = (gram *)realloc(p->g, sizeof(gram) * (size_t)p->cap); }
    key.text = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(len + 1));
    memcpy(key.text, t, sizeof(uint16_t) * (size_t)len);
    p->g[p->n++] = key;
}
static void profile_free(profile *p) {
    for (int32_t i = 0; i < p->n; ++i) free(p->g[i].text);
    free(p->g); p->g = NULL; p->n = p->cap = 0;
}
static int32_t profile_get(const profile *p, const gram *k) {            /* getOrDefault(key, 0) */
    for (int32_t i = 0; i < p->n; ++i) if (gram_cmp(&p->g[i], k) == 0) return p->g[i].count;
    return 0;
}

/* ShingleBased.getProfile: runs of [ \t\n\x0B\f\r] collapse to one space, then every k-gram */
static void ngram_profile(ustr s, int32_t n, int32_t k, profile *p) {
    uint16_t *t = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(n + 1));
    int32_t m = 0;
    for (int32_t i = 0; i < n;) {
        const uint16_t c = s[i];
        if (c == ' ' || (c >= 9 && c <= 13)) {
            t[m++] = ' ';
            while (i < n && (s[i] == ' ' || (s[i] >= 9 && s[i] <= 13))) ++i;
        } else { t[m++] = c; ++i; }
    }
    for (int32_t i = 0; i < m - k + 1; ++i) profile_add(p, t + i, k);
    free(t);
}

/* TokenBased.Tokenator (J/util/similarity/TokenBased.java:33-76) */
static const char *const ILLEGAL[] = {"the", "of", "and", "a", "an", "to", "in", "is", "you", "that", "it", "for",
                                      "on", "from", "are", "as", "with", "at", "or", "by", "but", "if"};
static int legal_token(const uint16_t *t, int32_t len) {
    if (len <= 1) return 0;
    for (size_t w = 0; w < sizeof(ILLEGAL) / sizeof(ILLEGAL[0]); ++w) {
        const size_t wl = strlen(ILLEGAL[w]);
        if ((size_t)len != wl) continue;
        int same = 1;
        for (size_t i = 0; i < wl; ++i) if (t[i] != (uint16_t)ILLEGAL[w][i]) { same = 0; break; }
        if (same) return 0;
    }
    return 1;
}
static void token_profile(ustr s, int32_t n, profile *p) {
    int32_t start = 0;
    for (int32_t pos = 0; pos < n; ++pos) {
        if (s[pos] == ' ' || pos == n - 1) {
            int32_t a = start, b = pos + 1;                /* substring(start, pos + 1).trim(): strips chars <= ' ' */
            while (a < b && s[a] <= ' ') ++a;
            while (b > a && s[b - 1] <= ' ') --b;
            if (legal_token(s + a, b - a)) profile_add(p, s + a, b - a);
            start = pos + 1;
        }
    }
}

static double profile_jaccard(const profile *a, const profile *b) {     /* PreComputed*Jaccard.similarity(profile, profile) */
    int32_t inter = 0;
    for (int32_t i = 0; i < a->n; ++i) if (profile_get(b, &a->g[i])) ++inter;
    const int32_t uni = a->n + b->n - inter;
    return (double)inter / (double)uni;                                   /* 0/0.0 = NaN, as in Java */
}
static double profile_norm(const profile *p) {
    double s = 0;
    for (int32_t i = 0; i < p->n; ++i) s += pow((double)p->g[i].count, 2);   /* exact: small integers */
    return sqrt(s);
}
static double profile_cosine(const profile *a, const profile *b) {      /* dotProduct / (norm * norm) */
    const profile *small = a->n > b->n ? b : a, *large = a->n > b->n ? a : b;
    int32_t dot = 0;                                                      /* mapToInt(...).sum(): int arithmetic */
    for (int32_t i = 0; i < small->n; ++i) dot = (int32_t)((uint32_t)dot + (uint32_t)(profile_get(large, &small->g[i]) * small->g[i].count));
    return (double)dot / (profile_norm(a) * profile_norm(b));
}

/* ---- Numeric (J/util/similarity/Numeric.java:19-44) ---- */
static int parse_int(ustr s, int32_t n, int32_t *out) {                  /* Integer.parseInt, ASCII digits */
    if (n <= 0) return 0;
    int32_t i = 0; int neg = 0;
    if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; if (n == 1) return 0; }
    int64_t v = 0;
    for (; i < n; ++i) {
        if (s[i] < '0' || s[i] > '9') return 0;
        v = v * 10 + (s[i] - '0');
        if (v > 2147483648LL) return 0;
    }
    if (!neg && v > 2147483647LL) return 0;
    *out = (int32_t)(neg ? -v : v);
    return 1;
}
/* returns the similarity; *threw = 1 when String.substring throws (the job dies, CompareJob's whole row is lost) */
static double sim_numeric(ustr s1, int32_t n1, ustr s2, int32_t n2, double alpha, double distance, int *threw) {
    if (n1 == 0 || n2 == 0) return 0;
    if (u_equal(s1, n1, s2, n2)) return 1;
    const int32_t s1hat = u_index_of(s1, n1, '^');
    const int32_t s2hat = u_index_of(s1, n1, '^');                        /* sic: the reference searches s1 twice (:32-33) */
    if (s1hat != -1) n1 = s1hat;
    if (s2hat != -1) { if (s2hat > n2) { *threw = 1; return 0; } n2 = s2hat; }
    int32_t a, b;
    if (!parse_int(s1, n1, &a) || !parse_int(s2, n2, &b)) return 0;       /* NumberFormatException -> 0 */
    int32_t diff = (int32_t)((uint32_t)a - (uint32_t)b);
    if (diff < 0 && diff != INT32_MIN) diff = -diff;                      /* Math.abs(int) */
    return pow(fabs((double)diff - distance) + 1, alpha - 1);
}

/* ---- Date (J/util/similarity/Date.java:30-65) ---- */
static int is_leap(int64_t y) { return (y % 4 == 0) && (y % 100 != 0 || y % 400 == 0); }
static int month_len(int64_t y, int m) {
    static const int L[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    return m == 2 && is_leap(y) ? 29 : L[m - 1];
}
static int64_t epoch_day(int64_t y, int m, int d) {                      /* LocalDate.toEpochDay */
    int64_t total = 365 * y;
    if (y >= 0) total += (y + 3) / 4 - (y + 99) / 100 + (y + 399) / 400;
    else        total -= y / -4 - y / -100 + y / -400;
    total += (367 * (int64_t)m - 362) / 12;
    total += d - 1;
    if (m > 2) { --total; if (!is_leap(y)) --total; }
    return total - 719528;
}
static int digits(ustr s, int32_t n, int32_t pos, int32_t count, int64_t *v) {
    if (pos + count > n) return 0;
    int64_t x = 0;
    for (int32_t i = 0; i < count; ++i) { if (s[pos + i] < '0' || s[pos + i] > '9') return 0; x = x * 10 + (s[pos + i] - '0'); }
    *v = x; return 1;
}
/* DateTimeFormatter.BASIC_ISO_DATE (STRICT): yyyyMMdd [offset +HHMMss | Z] */
static int parse_iso(ustr s, int32_t n, int64_t *y, int *m, int *d) {
    int64_t yy, mm, dd;
    if (!digits(s, n, 0, 4, &yy) || !digits(s, n, 4, 2, &mm) || !digits(s, n, 6, 2, &dd)) return 0;
    int32_t pos = 8;
    if (pos < n) {
        if (s[pos] == 'Z') ++pos;
        else if (s[pos] == '+' || s[pos] == '-') {
            int64_t h, mi = 0, se = 0;
            if (!digits(s, n, pos + 1, 2, &h)) return 0;
            pos += 3;
            if (digits(s, n, pos, 2, &mi)) { pos += 2; if (digits(s, n, pos, 2, &se)) pos += 2; }
            if (h > 18 || mi > 59 || se > 59 || (h == 18 && (mi || se))) return 0;
        }
        if (pos != n) return 0;
    }
    if (mm < 1 || mm > 12 || dd < 1 || dd > month_len(yy, (int)mm)) return 0;
    *y = yy; *m = (int)mm; *d = (int)dd;
    return 1;
}
/* DateTimeFormatter.ofPattern, the supported subset (SMART): letter runs yyyy|uuuu, MM|M, dd|d, quoted text, other
 * characters literally.  A year directly followed by fixed-width fields leaves them their digits (adjacent value
 * parsing).  Returns 0 on a mismatch (DateTimeParseException), -1 when the PATTERN is outside the subset. */
static int parse_pattern(const char *pat, ustr s, int32_t n, int64_t *y, int *m, int *d) {
    int32_t pos = 0;
    int64_t yy = 0, mm = 1, dd = 1; int have_y = 0, era_year = 0;
    for (size_t k = 0; pat[k];) {
        const char c = pat[k];
        if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) {
            size_t run = 1; while (pat[k + run] == c) ++run;
            int64_t v = 0;
            if ((c == 'y' || c == 'u') && run == 4) {
                /* digits reserved by directly following fixed-width fields */
                int32_t reserve = 0; size_t q = k + run;
                while (pat[q] == 'M' || pat[q] == 'd') { size_t r = 1; while (pat[q + r] == pat[q]) ++r; if (r != 2) break; reserve += 2; q += r; }
                int32_t avail = 0; while (pos + avail < n && s[pos + avail] >= '0' && s[pos + avail] <= '9') ++avail;
                int32_t take = avail - reserve; if (take > 9) take = 9;
                if (take < 4 || !digits(s, n, pos, take, &v)) return 0;
                pos += take; yy = v; have_y = 1; era_year = c == 'y';
            } else if ((c == 'M' || c == 'd') && run == 2) {
                if (!digits(s, n, pos, 2, &v)) return 0;
                pos += 2; if (c == 'M') mm = v; else dd = v;
            } else if ((c == 'M' || c == 'd') && run == 1) {
                int32_t avail = 0; while (pos + avail < n && avail < 9 && s[pos + avail] >= '0' && s[pos + avail] <= '9') ++avail;
                if (avail < 1 || !digits(s, n, pos, avail, &v)) return 0;
                pos += avail; if (c == 'M') mm = v; else dd = v;
            } else return -1;
            k += run;
        } else if (c == '\'') {
            size_t e = k + 1;
            if (pat[e] == '\'') { if (pos >= n || s[pos] != '\'') return 0; ++pos; k += 2; continue; }
            while (pat[e] && pat[e] != '\'') { if (pos >= n || s[pos] != (uint16_t)(unsigned char)pat[e]) return 0; ++pos; ++e; }
            if (!pat[e]) return -1;
            k = e + 1;
        } else {
            if (pos >= n || s[pos] != (uint16_t)(unsigned char)c) return 0;
            ++pos; ++k;
        }
    }
    if (pos != n || !have_y) return 0;
    if (era_year && yy < 1) return 0;
    if (mm < 1 || mm > 12 || dd < 1 || dd > 31) return 0;
    if (dd > month_len(yy, (int)mm)) dd = month_len(yy, (int)mm);           /* ResolverStyle.SMART clamps the day */
    *y = yy; *m = (int)mm; *d = (int)dd;
    return 1;
}
static double sim_date(ustr s1, int32_t n1, ustr s2, int32_t n2, const geo_sim_cfg *c) {
    if (n1 == 0 || n2 == 0) return 0;
    if (u_equal(s1, n1, s2, n2)) return 1;
    const int32_t h1 = u_index_of(s1, n1, '^'), h2 = u_index_of(s2, n2, '^');
    if (h1 != -1) n1 = h1;
    if (h2 != -1) n2 = h2;
    int64_t y1, y2; int m1, d1, m2, d2;
    const int iso = !c->pattern || strcmp(c->pattern, "iso") == 0;
    const int ok1 = iso ? parse_iso(s1, n1, &y1, &m1, &d1) : parse_pattern(c->pattern, s1, n1, &y1, &m1, &d1);
    if (ok1 <= 0) return 0;
    const int ok2 = iso ? parse_iso(s2, n2, &y2, &m2, &d2) : parse_pattern(c->pattern, s2, n2, &y2, &m2, &d2);
    if (ok2 <= 0) return 0;
    const int64_t e1 = epoch_day(y1, m1, d1), e2 = epoch_day(y2, m2, d2);
    if (c->time == GEO_TIME_BACKWARDS && e1 > e2) return 0;               /* d1.isAfter(d2)  */
    if (c->time == GEO_TIME_FORWARDS && e1 < e2) return 0;                /* d1.isBefore(d2) */
    int64_t between;
    if (c->method == GEO_SIM_DATE_DAYS) between = e2 - e1;
    else {
        const int64_t p1 = (y1 * 12 + (m1 - 1)) * 32 + d1, p2 = (y2 * 12 + (m2 - 1)) * 32 + d2;
        between = (p2 - p1) / 32;                                          /* LocalDate.monthsUntil */
        if (c->method == GEO_SIM_DATE_YEARS) between /= 12;
    }
    return pow(fabs(fabs((double)between) - c->distance) + 1, c->smooth - 1);
}

int geo_sim_pattern_supported(const char *pattern) {
    if (!pattern || strcmp(pattern, "iso") == 0) return 1;
    int have_y = 0;
    for (size_t k = 0; pattern[k];) {
        const char c = pattern[k];
        if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) {
            size_t run = 1; while (pattern[k + run] == c) ++run;
            if ((c == 'y' || c == 'u') && run == 4) have_y = 1;
            else if (!((c == 'M' || c == 'd') && run <= 2)) return 0;
            k += run;
        } else if (c == '\'') {
            size_t e = k + 1;
            if (pattern[e] == '\'') { k += 2; continue; }
            while (pattern[e] && pattern[e] != '\'') ++e;
            if (!pattern[e]) return 0;
            k = e + 1;
        } else ++k;
    }
    return have_y;
}

/* One metric.similarity(s1, s2) call (SimilarityGroup.toFunction picks the class). */
double geo_sim_pair(const geo_sim_cfg *c, const uint16_t *s1, int32_t n1, const uint16_t *s2, int32_t n2, int *threw) {
    int dummy = 0; if (!threw) threw = &dummy;
    *threw = 0;
    switch (c->method) {
    case GEO_SIM_JAROWINKLER: return geo_sim_jarowinkler(s1, n1, s2, n2);
    case GEO_SIM_LEVENSHTEIN: return sim_levenshtein(s1, n1, s2, n2);
    case GEO_SIM_NUMERIC:     return sim_numeric(s1, n1, s2, n2, c->smooth, c->distance, threw);
    case GEO_SIM_DATE_DAYS: case GEO_SIM_DATE_MONTHS: case GEO_SIM_DATE_YEARS: return sim_date(s1, n1, s2, n2, c);
    default: break;
    }
    if (u_equal(s1, n1, s2, n2)) return 1;
    if (c->method == GEO_SIM_NGRAM_COSINE && !(n1 >= c->ngram && n2 >= c->ngram)) return 0;   /* PreComputedNgramCosine.java:29-33 */
    profile a = {0}, b = {0};
    if (c->method == GEO_SIM_NGRAM_COSINE || c->method == GEO_SIM_NGRAM_JACCARD) { ngram_profile(s1, n1, c->ngram, &a); ngram_profile(s2, n2, c->ngram, &b); }
    else { token_profile(s1, n1, &a); token_profile(s2, n2, &b); }
    const double r = (c->method == GEO_SIM_NGRAM_JACCARD || c->method == GEO_SIM_TOKEN_JACCARD) ? profile_jaccard(&a, &b) : profile_cosine(&a, &b);
    profile_free(&a); profile_free(&b);
    return r;
}

/* The compare loop of Rdf2GrphConverter.convert (:127-186) for one CompareGroup with `threads: 1`: CompareJob i
 * walks target[startIndex..] (CompareJob.java:33-51), results arrive in job order.  src/tgt are positions into the
 * string table; *_vert the vertex ids (a job skips its own vertex).  Returns the number of pairs (which may
 * exceed cap; only cap are stored). */
int64_t geo_compare_group(const geo_sim_cfg *c, const int64_t *offset, const uint16_t *units,
                          const int32_t *src, const int32_t *src_vert, int32_t n_src,
                          const int32_t *tgt, const int32_t *tgt_vert, int32_t n_tgt, int upper_triangle,
                          int32_t *out_src, int32_t *out_tgt, float *out_sim, int64_t cap) {
    int64_t n = 0;
    for (int32_t i = 0; i < n_src; ++i) {
        const int64_t row_start = n;
        int dead = 0;
        const uint16_t *s1 = units + offset[src[i]]; const int32_t n1 = (int32_t)(offset[src[i] + 1] - offset[src[i]]);
        for (int32_t j = upper_triangle ? i + 1 : 0; j < n_tgt; ++j) {
            if (tgt_vert[j] == src_vert[i]) continue;
            const uint16_t *s2 = units + offset[tgt[j]]; const int32_t n2 = (int32_t)(offset[tgt[j] + 1] - offset[tgt[j]]);
            int threw = 0;
            const double sim = geo_sim_pair(c, s1, n1, s2, n2, &threw);
            if (threw) { dead = 1; break; }
            if (sim >= c->threshold) {
                if (n < cap) { out_src[n] = i; out_tgt[n] = j; out_sim[n] = (float)sim; }
                ++n;
            }
        }
        if (dead) n = row_start;                                          /* the job threw: ExecutionException, no result */
    }
    return n;
}
