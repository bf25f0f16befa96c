// similarity.hip -- literal-similarity edges on the device.
//
// Replaces the compare loop of Rdf2GrphConverter.convert (J/convert/Rdf2GrphConverter.java:127-186): for every
// CompareGroup, CompareJob i (J/compare/CompareJob.java:33-51) walks the target literals and keeps the pairs whose
// metric.similarity(s1, s2) >= threshold.  That is |source| x |target| string comparisons, the dominant cost of the
// reference's ingest for the shipped YAMLs (names by jarowinkler, titles by token / n-gram profiles).
//
// Layout: all labels of one group live in one pool of UTF-16 code units (java.lang.String semantics) plus
// (start, length) per string; profile metrics get, per string, a sorted list of (gram id, count) built on the host
// the way PreComputed.preCompute does at ingest.  One workgroup = one source literal x 256 targets, one lane = one
// pair; the source string / profile sits in LDS.  Integer and byte work throughout, bounded by LDS/L2 reads and
// divergent control flow, no MFMA.  Pairs that pass are appended through one atomic counter and sorted on the
// host into job order (i, then j), which is the order `threads: 1` produces.
//
// Exact metrics (jarowinkler, levenshtein, profiles) decide `>= threshold` on the device in the reference's own
// arithmetic (float for Jaro, IEEE double division / sqrt).  Numeric and Date* end in Math.pow: the device only
// FILTERS with a margin, the host recomputes every surviving pair with libm and applies the threshold.
#include "ge_common.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

constexpr int SIM_BLOCK = 256;
constexpr int MAX_UNITS = 1024;          // longest label (UTF-16 units) the string kernels take

struct StrRef { int32_t start, len; };
struct DateVal { int64_t epoch_day; int64_t packed_month; int32_t valid; int32_t pad; };

struct SimParams {
    const uint16_t *units; const StrRef *str;            // string table
    const int32_t *src, *src_vert, *tgt, *tgt_vert;     // positions into the string table + vertex ids
    int32_t n_src, n_tgt, src_begin, src_count;          // this launch covers sources [src_begin, src_begin+src_count)
    int32_t upper, method, ngram, time;
    double threshold, smooth, distance;
    // profiles
    const int32_t *prof_ptr; const int32_t *gram_id; const int32_t *gram_cnt; const double *prof_norm;
    const DateVal *date;
    const uint8_t *dead_row;                             // Numeric: jobs that die in String.substring
    // output
    int32_t *out_i, *out_j; double *out_sim; unsigned long long *counter; unsigned long long cap;
};

__device__ __forceinline__ bool units_equal(const uint16_t *a, const uint16_t *b, int n) {
    for (int k = 0; k < n; ++k) if (a[k] != b[k]) return false;
    return true;
}

// info.debatty JaroWinkler.similarity on (s1 in LDS, s2 in global).  W = 32-bit words of the match masks.
template <int W>
__device__ double jaro_winkler(const uint16_t *s1, int n1, const uint16_t *s2, int n2) {
    const uint16_t *mx = s1, *mn = s2; int nmx = n1, nmn = n2;
    if (!(n1 > n2)) { mx = s2; nmx = n2; mn = s1; nmn = n1; }
    int range = nmx / 2 - 1; if (range < 0) range = 0;
    uint32_t fmx[W], fmn[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { fmx[w] = 0; fmn[w] = 0; }
    int matches = 0;
    for (int mi = 0; mi < nmn; ++mi) {
        const uint16_t c1 = mn[mi];
        const int lo = max(mi - range, 0), hi = min(mi + range + 1, nmx);
        for (int xi = lo; xi < hi; ++xi) {
            if (!((fmx[xi >> 5] >> (xi & 31)) & 1u) && c1 == mx[xi]) {
                fmx[xi >> 5] |= 1u << (xi & 31); fmn[mi >> 5] |= 1u << (mi & 31); ++matches; break;
            }
        }
    }
    if (matches == 0) return 0.0;
    // k-th matched character of min against k-th matched character of max
    int transpositions = 0, pa = 0, pb = 0;
    for (int k = 0; k < matches; ++k) {
        while (!((fmn[pa >> 5] >> (pa & 31)) & 1u)) ++pa;
        while (!((fmx[pb >> 5] >> (pb & 31)) & 1u)) ++pb;
        transpositions += mn[pa] != mx[pb];
        ++pa; ++pb;
    }
    int prefix = 0;
    for (int k = 0; k < nmn; ++k) { if (s1[k] == s2[k]) ++prefix; else break; }
    const float m = (float)matches;
    const float jf = ((m / (float)n1 + m / (float)n2) + (m - (float)(transpositions / 2)) / m) / 3.0f;
    const double j = (double)jf;
    double jw = j;
    if (j > 0.7) {
        const double inv = 1.0 / (double)nmx;
        jw = j + (0.1 < inv ? 0.1 : inv) * (double)prefix * (1 - j);
    }
    return jw;
}

// Levenshtein.distance with the DP row over the shorter string; `give_up`: rows whose minimum exceeds it cannot
// reach the threshold any more (the distance never drops below a row minimum).
template <int L>
__device__ int levenshtein(const uint16_t *a, int na, const uint16_t *b, int nb, int give_up) {
    // a = outer (longer), b = inner (shorter)
    uint16_t row[L + 1];
    for (int j = 0; j <= nb; ++j) row[j] = (uint16_t)j;
    for (int i = 0; i < na; ++i) {
        const uint16_t ca = a[i];
        int diag = row[0], left = i + 1, rmin = left;
        row[0] = (uint16_t)left;
        for (int j = 0; j < nb; ++j) {
            const int up = row[j + 1];
            int best = min(left + 1, min(up + 1, diag + (ca != b[j])));
            row[j + 1] = (uint16_t)best;
            diag = up; left = best; rmin = min(rmin, best);
        }
        if (rmin > give_up) return rmin;
    }
    return row[nb];
}

__device__ bool dev_parse_int(const uint16_t *s, int n, int32_t *out) {          // Integer.parseInt (ASCII digits)
    if (n <= 0) return false;
    int i = 0; bool neg = false;
    if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; if (n == 1) return false; }
    long long v = 0;
    for (; i < n; ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
        if (v > 2147483648LL) return false;
    }
    if (!neg && v > 2147483647LL) return false;
    *out = (int32_t)(neg ? -v : v);
    return true;
}

template <int W>
__global__ __launch_bounds__(SIM_BLOCK) void k_similarity(SimParams p) {
    __shared__ uint16_t s_units[MAX_UNITS];
    __shared__ int32_t s_gid[MAX_UNITS], s_gcnt[MAX_UNITS];
    const int tiles = (p.n_tgt + SIM_BLOCK - 1) / SIM_BLOCK;
    const int i = p.src_begin + (int)(blockIdx.x / (unsigned)tiles);
    const int tile = (int)(blockIdx.x % (unsigned)tiles);
    const int j0 = tile * SIM_BLOCK;
    if (p.upper && j0 + SIM_BLOCK <= i + 1) return;                       // whole tile below the diagonal
    if (p.dead_row && p.dead_row[i]) return;
    const int sp = p.src[i];
    const StrRef r1 = p.str[sp];
    const bool profiles = p.method <= 3;
    int np1 = 0;
    if (profiles) {
        const int b = p.prof_ptr[sp]; np1 = p.prof_ptr[sp + 1] - b;
        for (int k = threadIdx.x; k < np1; k += SIM_BLOCK) { s_gid[k] = p.gram_id[b + k]; s_gcnt[k] = p.gram_cnt[b + k]; }
    }
    for (int k = threadIdx.x; k < r1.len; k += SIM_BLOCK) s_units[k] = p.units[r1.start + k];
    __syncthreads();
    const int j = j0 + (int)threadIdx.x;
    if (j >= p.n_tgt || (p.upper && j <= i)) return;
    if (p.tgt_vert[j] == p.src_vert[i]) return;                           // CompareJob.java:38
    const int tp = p.tgt[j];
    const StrRef r2 = p.str[tp];
    const uint16_t *s2 = p.units + r2.start;
    const int n1 = r1.len, n2 = r2.len;
    const bool same = n1 == n2 && units_equal(s_units, s2, n1);
    double sim;
    bool keep;
    switch (p.method) {
    case GE_SIM_JAROWINKLER: {
        if (same) { sim = 1.0; break; }
        // the most this pair can reach from its lengths alone: every character of the shorter string matched in order
        const int nmn = min(n1, n2), nmx = max(n1, n2);
        if (nmn == 0) { sim = 0.0; break; }
        const double jmax = ((double)nmn / n1 + (double)nmn / n2 + 1.0) / 3.0;
        const double wmax = jmax + fmin(0.1, 1.0 / nmx) * nmn * (1.0 - jmax);
        if (wmax < p.threshold - 1e-6) return;
        sim = jaro_winkler<W>(s_units, n1, s2, n2);
        break;
    }
    case GE_SIM_LEVENSHTEIN: {
        if (same) { sim = 1.0 - 0.0; break; }
        const int m_len = max(n1, n2);
        // sim >= threshold  <=>  d <= (1 - threshold) * m_len (up to rounding): one spare unit keeps the test exact below
        const double lim = (1.0 - p.threshold) * (double)m_len;
        const int give_up = lim >= (double)m_len ? m_len : (lim < 0 ? 0 : (int)lim + 1);
        if (abs(n1 - n2) > give_up) return;
        const int d = n1 >= n2 ? levenshtein<W * 32>(s_units, n1, s2, n2, give_up) : levenshtein<W * 32>(s2, n2, s_units, n1, give_up);
        sim = 1.0 - (double)d / (double)m_len;
        break;
    }
    case GE_SIM_NUMERIC: {
        if (n1 == 0 || n2 == 0) { sim = 0.0; break; }
        if (same) { sim = 1.0; break; }
        int hat = -1;
        for (int k = 0; k < n1; ++k) if (s_units[k] == '^') { hat = k; break; }
        const int l1 = hat != -1 ? hat : n1;
        const int l2 = hat != -1 ? hat : n2;                               // sic: Numeric.java:33 takes s1's position for s2 (rows where
        int32_t a, b;                                                      // that exceeds a target's length are in dead_row)
        if (l2 > n2 || !dev_parse_int(s_units, l1, &a) || !dev_parse_int(s2, l2, &b)) { sim = 0.0; break; }
        int32_t diff = (int32_t)((uint32_t)a - (uint32_t)b);
        if (diff < 0 && diff != INT32_MIN) diff = -diff;
        sim = pow(fabs((double)diff - p.distance) + 1, p.smooth - 1);
        if (!(sim >= p.threshold - 1e-9 * fmax(1.0, fabs(p.threshold)))) return;       // margin: the host decides with libm
        goto emit;
    }
    case GE_SIM_DATE_DAYS: case GE_SIM_DATE_MONTHS: case GE_SIM_DATE_YEARS: {
        if (n1 == 0 || n2 == 0) { sim = 0.0; break; }
        if (same) { sim = 1.0; break; }
        const DateVal d1 = p.date[sp], d2 = p.date[tp];
        if (!d1.valid || !d2.valid) { sim = 0.0; break; }
        if (p.time == GE_TIME_BACKWARDS && d1.epoch_day > d2.epoch_day) { sim = 0.0; break; }
        if (p.time == GE_TIME_FORWARDS && d1.epoch_day < d2.epoch_day) { sim = 0.0; break; }
        long long between;
        if (p.method == GE_SIM_DATE_DAYS) between = d2.epoch_day - d1.epoch_day;
        else { between = (d2.packed_month - d1.packed_month) / 32; if (p.method == GE_SIM_DATE_YEARS) between /= 12; }
        sim = pow(fabs(fabs((double)between) - p.distance) + 1, p.smooth - 1);
        if (!(sim >= p.threshold - 1e-9 * fmax(1.0, fabs(p.threshold)))) return;
        goto emit;
    }
    default: {                                                             // profile metrics
        if (same) { sim = 1.0; break; }
        if (p.method == GE_SIM_NGRAM_COSINE && !(n1 >= p.ngram && n2 >= p.ngram)) { sim = 0.0; break; }
        const int b2 = p.prof_ptr[tp], np2 = p.prof_ptr[tp + 1] - b2;
        const int32_t *g2 = p.gram_id + b2, *c2 = p.gram_cnt + b2;
        int a = 0, b = 0, inter = 0; uint32_t dot = 0;
        while (a < np1 && b < np2) {                                       // both lists ascend by gram id
            const int ga = s_gid[a], gb = g2[b];
            if (ga == gb) { ++inter; dot += (uint32_t)(s_gcnt[a] * c2[b]); ++a; ++b; }
            else if (ga < gb) ++a; else ++b;
        }
        if (p.method == GE_SIM_NGRAM_JACCARD || p.method == GE_SIM_TOKEN_JACCARD) sim = (double)inter / (double)(np1 + np2 - inter);
        else sim = (double)(int32_t)dot / (p.prof_norm[sp] * p.prof_norm[tp]);
        break;
    }
    }
    keep = sim >= p.threshold;                                             // NaN (empty profiles) compares false, as in Java
    if (!keep) return;
emit:
    {
        const unsigned long long slot = atomicAdd(p.counter, 1ULL);
        if (slot < p.cap) { p.out_i[slot] = i; p.out_j[slot] = j; p.out_sim[slot] = sim; }
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side: profiles (PreComputed.preCompute), date parsing, the exact Numeric/Date recomputation
// ---------------------------------------------------------------------------------------------------------
using u16 = std::u16string;

bool is_space_class(char16_t c) { return c == u' ' || (c >= 9 && c <= 13); }       // java.util.regex \s

// ShingleBased.getProfile(string): whitespace runs -> one space, every k-gram counted
void ngram_grams(const char16_t *s, int n, int k, std::vector<u16> &out) {
    u16 t; t.reserve((size_t)n);
    for (int i = 0; i < n;) {
        if (is_space_class(s[i])) { t.push_back(u' '); while (i < n && is_space_class(s[i])) ++i; }
        else t.push_back(s[i++]);
    }
    for (int i = 0; i + k <= (int)t.size(); ++i) out.emplace_back(t, (size_t)i, (size_t)k);
}

// TokenBased.Tokenator (J/util/similarity/TokenBased.java:33-76)
void token_grams(const char16_t *s, int n, std::vector<u16> &out) {
    static const char16_t *const stop[] = {u"the", u"of", u"and", u"a", u"an", u"to", u"in", u"is", u"you", u"that", u"it", u"for",
                                           u"on", u"from", u"are", u"as", u"with", u"at", u"or", u"by", u"but", u"if"};
    int begin = 0;
    for (int pos = 0; pos < n; ++pos) {
        if (s[pos] != u' ' && pos != n - 1) continue;
        int a = begin, b = pos + 1;
        while (a < b && s[a] <= u' ') ++a;                       // String.trim
        while (b > a && s[b - 1] <= u' ') --b;
        begin = pos + 1;
        if (b - a <= 1) continue;
        u16 tok(s + a, (size_t)(b - a));
        bool illegal = false;
        for (const char16_t *w : stop) if (tok == w) { illegal = true; break; }
        if (!illegal) out.push_back(std::move(tok));
    }
}

bool leap(int64_t y) { return (y % 4 == 0) && (y % 100 != 0 || y % 400 == 0); }
int month_length(int64_t y, int m) {
    static const int L[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    return (m == 2 && leap(y)) ? 29 : L[m - 1];
}
int64_t to_epoch_day(int64_t y, int m, int d) {                  // java.time.LocalDate.toEpochDay
    int64_t t = 365 * y;
    t += y >= 0 ? (y + 3) / 4 - (y + 99) / 100 + (y + 399) / 400 : -(y / -4 - y / -100 + y / -400);
    t += (367 * (int64_t)m - 362) / 12 + (d - 1);
    if (m > 2) t -= leap(y) ? 1 : 2;
    return t - 719528;
}
bool take_digits(const char16_t *s, int n, int pos, int count, int64_t &v) {
    if (count <= 0 || pos + count > n) return false;
    v = 0;
    for (int k = 0; k < count; ++k) { const char16_t c = s[pos + k]; if (c < u'0' || c > u'9') return false; v = v * 10 + (c - u'0'); }
    return true;
}

// One compiled date format: "iso" (DateTimeFormatter.BASIC_ISO_DATE, STRICT) or the supported subset of
// DateTimeFormatter.ofPattern (SMART): yyyy|uuuu, MM|M, dd|d, quoted text, other characters literally.
struct DateFormat {
    struct Item { char kind; int width; u16 text; int reserve; };      // kind: 'y','u','M','d','L'(iteral)
    bool iso = true;
    std::vector<Item> items;

    static bool compile(const char *pattern, DateFormat &f) {
        f.items.clear();
        f.iso = !pattern || std::strcmp(pattern, "iso") == 0;
        if (f.iso) return true;
        bool year = false;
        for (size_t k = 0; pattern[k];) {
            const char c = pattern[k];
            if (std::isalpha((unsigned char)c)) {
                size_t run = 1; while (pattern[k + run] == c) ++run;
                if ((c == 'y' || c == 'u') && run == 4) { f.items.push_back({c, 4, u16(), 0}); year = true; }
                else if ((c == 'M' || c == 'd') && run <= 2) f.items.push_back({c, (int)run, u16(), 0});
                else return false;
                k += run;
            } else if (c == '\'') {
                if (pattern[k + 1] == '\'') { f.items.push_back({'L', 0, u16(1, u'\''), 0}); k += 2; continue; }
                size_t e = k + 1; u16 lit;
                while (pattern[e] && pattern[e] != '\'') lit.push_back((char16_t)(unsigned char)pattern[e++]);
                if (!pattern[e]) return false;
                f.items.push_back({'L', 0, lit, 0});
                k = e + 1;
            } else { f.items.push_back({'L', 0, u16(1, (char16_t)(unsigned char)c), 0}); ++k; }
        }
        // adjacent value parsing: a year leaves the fixed-width fields right behind it their digits
        for (size_t q = 0; q < f.items.size(); ++q)
            if (f.items[q].kind == 'y' || f.items[q].kind == 'u')
                for (size_t r = q + 1; r < f.items.size() && (f.items[r].kind == 'M' || f.items[r].kind == 'd') && f.items[r].width == 2; ++r)
                    f.items[q].reserve += 2;
        return year;
    }

    bool parse(const char16_t *s, int n, int64_t &y, int &m, int &d) const {
        int64_t yy = 0, mm = 1, dd = 1;
        int pos = 0;
        if (iso) {
            if (!take_digits(s, n, 0, 4, yy) || !take_digits(s, n, 4, 2, mm) || !take_digits(s, n, 6, 2, dd)) return false;
            pos = 8;
            if (pos < n) {
                if (s[pos] == u'Z') ++pos;
                else if (s[pos] == u'+' || s[pos] == u'-') {
                    int64_t h = 0, mi = 0, se = 0;
                    if (!take_digits(s, n, pos + 1, 2, h)) return false;
                    pos += 3;
                    if (take_digits(s, n, pos, 2, mi)) { pos += 2; if (take_digits(s, n, pos, 2, se)) pos += 2; }
                    if (h > 18 || mi > 59 || se > 59 || (h == 18 && (mi || se))) return false;
                }
                if (pos != n) return false;
            }
            if (mm < 1 || mm > 12 || dd < 1 || dd > month_length(yy, (int)mm)) return false;          // STRICT
            y = yy; m = (int)mm; d = (int)dd;
            return true;
        }
        bool era_year = false;
        for (const Item &it : items) {
            int64_t v = 0;
            if (it.kind == 'L') {
                for (char16_t c : it.text) { if (pos >= n || s[pos] != c) return false; ++pos; }
            } else if (it.kind == 'y' || it.kind == 'u') {
                int avail = 0; while (pos + avail < n && s[pos + avail] >= u'0' && s[pos + avail] <= u'9') ++avail;
                const int take = std::min(avail - it.reserve, 9);
                if (take < 4 || !take_digits(s, n, pos, take, v)) return false;
                pos += take; yy = v; era_year = it.kind == 'y';
            } else {
                int take = 2;
                if (it.width == 1) { take = 0; while (pos + take < n && take < 9 && s[pos + take] >= u'0' && s[pos + take] <= u'9') ++take; }
                if (!take_digits(s, n, pos, take, v)) return false;
                pos += take; (it.kind == 'M' ? mm : dd) = v;
            }
        }
        if (pos != n) return false;
        if (era_year && yy < 1) return false;
        if (mm < 1 || mm > 12 || dd < 1 || dd > 31) return false;
        y = yy; m = (int)mm; d = (int)std::min<int64_t>(dd, month_length(yy, (int)mm));               // SMART clamps the day
        return true;
    }
};

int index_of(const char16_t *s, int n, char16_t c) { for (int k = 0; k < n; ++k) if (s[k] == c) return k; return -1; }

bool host_parse_int(const char16_t *s, int n, int32_t &out) {
    if (n <= 0) return false;
    int i = 0; bool neg = false;
    if (s[0] == u'-' || s[0] == u'+') { neg = s[0] == u'-'; i = 1; if (n == 1) return false; }
    int64_t v = 0;
    for (; i < n; ++i) {
        if (s[i] < u'0' || s[i] > u'9') return false;
        v = v * 10 + (s[i] - u'0');
        if (v > 2147483648LL) return false;
    }
    if (!neg && v > 2147483647LL) return false;
    out = (int32_t)(neg ? -v : v);
    return true;
}

// Numeric.similarity for a pair that did not throw (J/util/similarity/Numeric.java:19-44)
double host_numeric(const char16_t *s1, int n1, const char16_t *s2, int n2, double alpha, double distance) {
    if (n1 == 0 || n2 == 0) return 0;
    if (n1 == n2 && std::equal(s1, s1 + n1, s2)) return 1;
    const int hat = index_of(s1, n1, u'^');
    if (hat != -1) { if (hat > n2) return 0; n1 = hat; n2 = hat; }     // both cuts use s1's position (:32-36); hat > n2 throws (dead row)
    int32_t a, b;
    if (!host_parse_int(s1, n1, a) || !host_parse_int(s2, n2, b)) return 0;
    int32_t diff = (int32_t)((uint32_t)a - (uint32_t)b);
    if (diff < 0 && diff != INT32_MIN) diff = -diff;
    return std::pow(std::fabs((double)diff - distance) + 1, alpha - 1);
}

}  // namespace

struct ge_sim_pairs {
    std::vector<int32_t> src, tgt;
    std::vector<float> sim;
};

namespace {

template <typename T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    ge_status alloc(size_t n) { GE_HIP(hipMalloc((void **)&p, sizeof(T) * std::max<size_t>(n, 1))); return GE_OK; }
    ge_status upload(const std::vector<T> &v) {
        if (ge_status s = alloc(v.size())) return s;
        if (!v.empty()) GE_HIP(hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
        return GE_OK;
    }
};

void launch(int words, unsigned blocks, const SimParams &p) {
    if (words <= 2)      hipLaunchKernelGGL((k_similarity<2>),  dim3(blocks), dim3(SIM_BLOCK), 0, 0, p);
    else if (words <= 8) hipLaunchKernelGGL((k_similarity<8>),  dim3(blocks), dim3(SIM_BLOCK), 0, 0, p);
    else                 hipLaunchKernelGGL((k_similarity<32>), dim3(blocks), dim3(SIM_BLOCK), 0, 0, p);
}

}  // namespace

extern "C" {

void ge_sim_cfg_default(ge_sim_cfg *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->method = GE_SIM_JAROWINKLER;
    cfg->ngram = 3;                       // SimilarityGroup.getNgram: 0 -> 3
    cfg->smooth = 1.0;                    // getSmooth: 0 -> 1
    cfg->time = GE_TIME_BIDIRECTIONAL;    // getTime: null -> bidirectional
    cfg->pattern = nullptr;               // getPattern: null -> "iso"
}

int32_t ge_sim_cfg_size(void) { return (int32_t)sizeof(ge_sim_cfg); }

int32_t ge_sim_pattern_supported(const char *pattern) {
    DateFormat f;
    return DateFormat::compile(pattern, f) ? 1 : 0;
}

static ge_status similarity_pairs_impl(const ge_strings *strings, const int32_t *src, const int32_t *src_vert, int32_t n_src,
                                       const int32_t *tgt, const int32_t *tgt_vert, int32_t n_tgt,
                                       const ge_sim_cfg *cfg, ge_sim_pairs **result) {
    if (!strings || !cfg || !result) return ge::fail(GE_ERR_ARG, "ge_similarity_pairs: null argument");
    *result = nullptr;
    if (n_src < 0 || n_tgt < 0 || strings->count < 0) return ge::fail(GE_ERR_ARG, "ge_similarity_pairs: negative count");
    if ((n_src && (!src || !src_vert)) || (n_tgt && (!tgt || !tgt_vert))) return ge::fail(GE_ERR_ARG, "ge_similarity_pairs: null index array");
    if (strings->count && (!strings->offset || (!strings->units && strings->offset[strings->count] > 0)))
        return ge::fail(GE_ERR_ARG, "ge_similarity_pairs: null string table");
    if (cfg->method < GE_SIM_NGRAM_COSINE || cfg->method > GE_SIM_DATE_YEARS) return ge::fail(GE_ERR_ARG, "unknown similarity method %d", cfg->method);
    if (cfg->time < GE_TIME_BACKWARDS || cfg->time > GE_TIME_BIDIRECTIONAL) return ge::fail(GE_ERR_ARG, "unknown time direction %d", cfg->time);
    const bool profiles = cfg->method <= GE_SIM_TOKEN_JACCARD;
    const bool ngrams = cfg->method == GE_SIM_NGRAM_COSINE || cfg->method == GE_SIM_NGRAM_JACCARD;
    const bool dates = cfg->method >= GE_SIM_DATE_DAYS;
    const int ngram = cfg->ngram == 0 ? 3 : cfg->ngram;
    if (ngrams && ngram <= 0) return ge::fail(GE_ERR_ARG, "k should be positive!");                  // ShingleBased ctor
    const double smooth = cfg->smooth == 0 ? 1.0 : cfg->smooth;
    if (cfg->upper_triangle && n_src != n_tgt) return ge::fail(GE_ERR_ARG, "upper_triangle needs source == target");
    int32_t job_begin = 0, job_end = n_src;
    if (cfg->job_begin != 0 || cfg->job_end != 0) {
        if (cfg->job_begin < 0 || cfg->job_end < cfg->job_begin || cfg->job_end > n_src) return ge::fail(GE_ERR_ARG, "job range [%d,%d) outside the %d jobs", cfg->job_begin, cfg->job_end, n_src);
        job_begin = cfg->job_begin; job_end = cfg->job_end;
    }
    DateFormat fmt;
    if (dates && !DateFormat::compile(cfg->pattern, fmt))
        return ge::fail(GE_ERR_ARG, "date pattern '%s' is outside the supported subset (iso, or yyyy/uuuu MM/M dd/d with literals)", cfg->pattern ? cfg->pattern : "");
    const int32_t S = strings->count;
    for (int32_t k = 0; k < n_src; ++k) if (src[k] < 0 || src[k] >= S) return ge::fail(GE_ERR_ARG, "source[%d] = %d outside the string table", k, src[k]);
    for (int32_t k = 0; k < n_tgt; ++k) if (tgt[k] < 0 || tgt[k] >= S) return ge::fail(GE_ERR_ARG, "target[%d] = %d outside the string table", k, tgt[k]);
    if (S && strings->offset[S] >= (int64_t)1 << 31) return ge::fail(GE_ERR_ARG, "string table larger than 2^31 code units");

    std::unique_ptr<ge_sim_pairs> holder(new ge_sim_pairs());
    ge_sim_pairs *out = holder.get();
    if (n_src == 0 || n_tgt == 0 || job_begin == job_end) { *result = holder.release(); return GE_OK; }
    if (ge_status st = ge::select_device(cfg->device)) return st;

    // ---- string table ------------------------------------------------------------------------------------
    const char16_t *U = reinterpret_cast<const char16_t *>(strings->units);
    std::vector<StrRef> refs((size_t)S);
    std::vector<uint8_t> used((size_t)S, 0);
    for (int32_t k = 0; k < n_src; ++k) used[(size_t)src[k]] = 1;
    for (int32_t k = 0; k < n_tgt; ++k) used[(size_t)tgt[k]] = 1;
    int max_len = 0;
    for (int32_t s = 0; s < S; ++s) {
        const int64_t b = strings->offset[s], e = strings->offset[s + 1];
        if (e < b) { return ge::fail(GE_ERR_ARG, "string offsets must ascend"); }
        refs[(size_t)s] = {(int32_t)b, (int32_t)(e - b)};
        if (used[(size_t)s]) max_len = std::max<int>(max_len, (int)(e - b));
    }
    const bool string_kernel = cfg->method == GE_SIM_JAROWINKLER || cfg->method == GE_SIM_LEVENSHTEIN;
    if (max_len > MAX_UNITS) {
        return ge::fail(GE_ERR_ARG, "a label of %d UTF-16 units exceeds the %d the similarity kernels take", max_len, MAX_UNITS);
    }

    // ---- profiles (PreComputed.preCompute at ingest) ------------------------------------------------------
    std::vector<int32_t> prof_ptr, gram_id, gram_cnt; std::vector<double> prof_norm;
    if (profiles) {
        std::unordered_map<u16, int32_t> dict;
        prof_ptr.assign((size_t)S + 1, 0); prof_norm.assign((size_t)S, 0.0);
        std::vector<u16> grams; std::vector<int32_t> ids;
        for (int32_t s = 0; s < S; ++s) {
            prof_ptr[(size_t)s] = (int32_t)gram_id.size();
            if (!used[(size_t)s]) continue;
            grams.clear(); ids.clear();
            if (ngrams) ngram_grams(U + refs[(size_t)s].start, refs[(size_t)s].len, ngram, grams);
            else token_grams(U + refs[(size_t)s].start, refs[(size_t)s].len, grams);
            for (auto &g : grams) ids.push_back(dict.emplace(g, (int32_t)dict.size()).first->second);
            std::sort(ids.begin(), ids.end());
            double sq = 0;
            for (size_t a = 0; a < ids.size();) {
                size_t b = a; while (b < ids.size() && ids[b] == ids[a]) ++b;
                gram_id.push_back(ids[a]); gram_cnt.push_back((int32_t)(b - a));
                sq += (double)(b - a) * (double)(b - a);                    // Math.pow(v, 2), exact
                a = b;
            }
            prof_norm[(size_t)s] = std::sqrt(sq);
            if ((int)(gram_id.size() - (size_t)prof_ptr[(size_t)s]) > MAX_UNITS) { return ge::fail(GE_ERR_ARG, "profile with more than %d distinct grams", MAX_UNITS); }
        }
        prof_ptr[(size_t)S] = (int32_t)gram_id.size();
    }
    // ---- dates -------------------------------------------------------------------------------------------
    std::vector<DateVal> date;
    if (dates) {
        date.assign((size_t)S, DateVal{0, 0, 0, 0});
        for (int32_t s = 0; s < S; ++s) {
            if (!used[(size_t)s]) continue;
            const char16_t *p = U + refs[(size_t)s].start; int n = refs[(size_t)s].len;
            const int hat = index_of(p, n, u'^');
            if (hat != -1) n = hat;
            int64_t y; int m, d;
            if (fmt.parse(p, n, y, m, d)) date[(size_t)s] = {to_epoch_day(y, m, d), (y * 12 + (m - 1)) * 32 + d, 1, 0};
        }
    }
    // ---- Numeric: jobs that die in s2.substring(0, s1hat) (Numeric.java:36) --------------------------------
    std::vector<uint8_t> dead;
    if (cfg->method == GE_SIM_NUMERIC) {
        dead.assign((size_t)n_src, 0);
        // shortest non-empty target at or after each position
        std::vector<int32_t> suffix_min((size_t)n_tgt + 1, INT32_MAX);
        for (int32_t j = n_tgt - 1; j >= 0; --j) {
            const int32_t len = refs[(size_t)tgt[j]].len;
            suffix_min[(size_t)j] = std::min(suffix_min[(size_t)j + 1], len > 0 ? len : INT32_MAX);
        }
        for (int32_t i = 0; i < n_src; ++i) {
            const StrRef r = refs[(size_t)src[i]];
            const int hat = index_of(U + r.start, r.len, u'^');
            if (r.len == 0 || hat == -1) continue;
            const int32_t from = cfg->upper_triangle ? i + 1 : 0;
            // a shorter non-empty target differs from s1 and is another vertex, so the job reaches the substring
            if (from <= n_tgt && suffix_min[(size_t)std::min(from, n_tgt)] < hat) dead[(size_t)i] = 1;
        }
    }

    // ---- device --------------------------------------------------------------------------------------------
    DevBuf<uint16_t> d_units; DevBuf<StrRef> d_refs; DevBuf<int32_t> d_src, d_srcv, d_tgt, d_tgtv, d_pptr, d_gid, d_gcnt;
    DevBuf<double> d_norm; DevBuf<DateVal> d_date; DevBuf<uint8_t> d_dead; DevBuf<unsigned long long> d_counter;
    ge_status st = GE_OK;
    auto fail = [&](ge_status s) { return s; };
    {
        const size_t total = (size_t)strings->offset[S];
        if ((st = d_units.alloc(total))) return fail(st);
        if (total) { hipError_t e = hipMemcpy(d_units.p, strings->units, sizeof(uint16_t) * total, hipMemcpyHostToDevice); if (e != hipSuccess) return fail(ge::fail(GE_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e))); }
    }
    if ((st = d_refs.upload(refs))) return fail(st);
    if ((st = d_src.upload(std::vector<int32_t>(src, src + n_src)))) return fail(st);
    if ((st = d_srcv.upload(std::vector<int32_t>(src_vert, src_vert + n_src)))) return fail(st);
    if ((st = d_tgt.upload(std::vector<int32_t>(tgt, tgt + n_tgt)))) return fail(st);
    if ((st = d_tgtv.upload(std::vector<int32_t>(tgt_vert, tgt_vert + n_tgt)))) return fail(st);
    if (profiles) {
        if ((st = d_pptr.upload(prof_ptr)) || (st = d_gid.upload(gram_id)) || (st = d_gcnt.upload(gram_cnt)) || (st = d_norm.upload(prof_norm))) return fail(st);
    }
    if (dates && (st = d_date.upload(date))) return fail(st);
    if (!dead.empty() && (st = d_dead.upload(dead))) return fail(st);
    if ((st = d_counter.alloc(1))) return fail(st);

    SimParams p{};
    p.units = d_units.p; p.str = d_refs.p; p.src = d_src.p; p.src_vert = d_srcv.p; p.tgt = d_tgt.p; p.tgt_vert = d_tgtv.p;
    p.n_src = n_src; p.n_tgt = n_tgt; p.upper = cfg->upper_triangle ? 1 : 0; p.method = cfg->method; p.ngram = ngram; p.time = cfg->time;
    p.threshold = cfg->threshold; p.smooth = smooth; p.distance = cfg->distance;
    p.prof_ptr = d_pptr.p; p.gram_id = d_gid.p; p.gram_cnt = d_gcnt.p; p.prof_norm = d_norm.p; p.date = d_date.p;
    p.dead_row = dead.empty() ? nullptr : d_dead.p;
    p.counter = d_counter.p;

    const int words = string_kernel ? (max_len + 31) / 32 : 1;
    const int64_t tiles = ((int64_t)n_tgt + SIM_BLOCK - 1) / SIM_BLOCK;
    const int32_t rows_per_launch = (int32_t)std::max<int64_t>(1, std::min<int64_t>(job_end - job_begin, ((int64_t)1 << 30) / tiles));
    size_t cap = (size_t)1 << 22;                                        // pairs per launch before the buffers grow
    DevBuf<int32_t> d_oi, d_oj; DevBuf<double> d_os;
    if ((st = d_oi.alloc(cap)) || (st = d_oj.alloc(cap)) || (st = d_os.alloc(cap))) return fail(st);
    struct Hit { int32_t i, j; double sim; };
    std::vector<Hit> hits;
    std::vector<int32_t> hi, hj; std::vector<double> hs;
    for (int32_t begin = job_begin; begin < job_end;) {
        const int32_t count = std::min(rows_per_launch, job_end - begin);
        p.src_begin = begin; p.src_count = count; p.out_i = d_oi.p; p.out_j = d_oj.p; p.out_sim = d_os.p; p.cap = cap;
        hipError_t e = hipMemset(d_counter.p, 0, sizeof(unsigned long long));
        if (e != hipSuccess) return fail(ge::fail(GE_ERR_HIP, "hipMemset: %s", hipGetErrorString(e)));
        launch(words, (unsigned)((int64_t)count * tiles), p);
        unsigned long long found = 0;
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(&found, d_counter.p, sizeof(found), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(ge::fail(GE_ERR_HIP, "similarity kernel: %s", hipGetErrorString(e)));
        if (found > cap) {                                                // did not fit: enlarge and repeat these rows
            cap = (size_t)found + (found >> 3);
            DevBuf<int32_t> ni, nj; DevBuf<double> ns;
            if ((st = ni.alloc(cap)) || (st = nj.alloc(cap)) || (st = ns.alloc(cap))) return fail(st);
            std::swap(d_oi.p, ni.p); std::swap(d_oj.p, nj.p); std::swap(d_os.p, ns.p);
            continue;
        }
        hi.resize((size_t)found); hj.resize((size_t)found); hs.resize((size_t)found);
        if (found) {
            e = hipMemcpy(hi.data(), d_oi.p, sizeof(int32_t) * found, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(hj.data(), d_oj.p, sizeof(int32_t) * found, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(hs.data(), d_os.p, sizeof(double) * found, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return fail(ge::fail(GE_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e)));
        }
        for (size_t k = 0; k < (size_t)found; ++k) hits.push_back({hi[k], hj[k], hs[k]});
        begin += count;
    }
    // job order: CompareJob i in submission order, j ascending inside it (threads: 1)
    std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.i != b.i ? a.i < b.i : a.j < b.j; });
    const bool refine = cfg->method == GE_SIM_NUMERIC || dates;
    for (const Hit &h : hits) {
        double sim = h.sim;
        if (refine) {                                                     // Math.pow on the host decides
            const StrRef a = refs[(size_t)src[h.i]], b = refs[(size_t)tgt[h.j]];
            if (cfg->method == GE_SIM_NUMERIC) sim = host_numeric(U + a.start, a.len, U + b.start, b.len, smooth, cfg->distance);
            else {                                                        // Date.similarity (J/util/similarity/Date.java:30-65)
                const DateVal d1 = date[(size_t)src[h.i]], d2 = date[(size_t)tgt[h.j]];
                if (a.len == 0 || b.len == 0) sim = 0;
                else if (a.len == b.len && std::equal(U + a.start, U + a.start + a.len, U + b.start)) sim = 1;
                else if (!d1.valid || !d2.valid) sim = 0;                 // DateTimeParseException
                else if (cfg->time == GE_TIME_BACKWARDS && d1.epoch_day > d2.epoch_day) sim = 0;
                else if (cfg->time == GE_TIME_FORWARDS && d1.epoch_day < d2.epoch_day) sim = 0;
                else {
                    int64_t between = cfg->method == GE_SIM_DATE_DAYS ? d2.epoch_day - d1.epoch_day : (d2.packed_month - d1.packed_month) / 32;
                    if (cfg->method == GE_SIM_DATE_YEARS) between /= 12;
                    sim = std::pow(std::fabs(std::fabs((double)between) - cfg->distance) + 1, smooth - 1);
                }
            }
            if (!(sim >= cfg->threshold)) continue;
        }
        out->src.push_back(h.i); out->tgt.push_back(h.j); out->sim.push_back((float)sim);
    }
    *result = holder.release();
    return GE_OK;
}

ge_status ge_sim_pairs_get(const ge_sim_pairs *r, int64_t *count, const int32_t **src_pos, const int32_t **tgt_pos, const float **similarity) {
    if (!r) return ge::fail(GE_ERR_ARG, "ge_sim_pairs_get: null handle");
    if (count) *count = (int64_t)r->src.size();
    if (src_pos) *src_pos = r->src.data();
    if (tgt_pos) *tgt_pos = r->tgt.data();
    if (similarity) *similarity = r->sim.data();
    return GE_OK;
}

void ge_sim_pairs_destroy(ge_sim_pairs *r) { delete r; }

ge_status ge_similarity_pairs(const ge_strings *strings, const int32_t *source, const int32_t *source_vertex, int32_t n_source,
                              const int32_t *target, const int32_t *target_vertex, int32_t n_target,
                              const ge_sim_cfg *cfg, ge_sim_pairs **result) {
    GE_GUARD(similarity_pairs_impl(strings, source, source_vertex, n_source, target, target_vertex, n_target, cfg, result));
}

}  // extern "C"
