// host_rng.cpp -- the glibc rand() stream and the position coder.
//
// srand/rand are part of the reference's contract (E:1386, 1199, 1336): the TYPE_3 additive
// feedback generator of glibc.  random_r on a private 128-byte state is that same generator
// (rand() is random() on glibc's default 128-byte table), without touching the process-wide
// stream of whoever loaded this library.
#include <cstdarg>
#include <cstring>
#include "lhgt_common.hpp"

namespace lhgt {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

int rng_next(lhgt_ctx* ctx) {
    int32_t r = 0;
    random_r((struct random_data*)ctx->rng, &r);
    return (int)r;
}

int build_hash_params(const int16_t* cc, int k, int e, HashParams* hp) {
    memset(hp, 0, sizeof *hp);
    hp->k = k;
    hp->e = e;
    for (int z = 0; z < k; z++)
        for (int i = 0; i < e; i++) {
            int m = cc[z * e + i];
            if (m < 0 || m > 2) {
                set_error("coder entry [%d][%d] = %d is not one of 0,1,2 (corrupt index header?)", z, i, m);
                return LHGT_E_FORMAT;
            }
            hp->mask[i][m] |= 1u << (k - 1 - z);
        }
    return LHGT_OK;
}

}  // namespace lhgt

using namespace lhgt;

extern "C" {

const char* lhgt_last_error(void) { return lhgt::last_error(); }
int lhgt_abi_version(void) { return LHGT_ABI_VERSION; }

int lhgt_rng_seed(lhgt_ctx* ctx, unsigned seed) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    sampling_join(ctx);
    if (!ctx->rng) ctx->rng = calloc(1, sizeof(struct random_data));
    memset(ctx->rng, 0, sizeof(struct random_data));
    memset(ctx->rng_state, 0, sizeof ctx->rng_state);
    if (initstate_r(seed, ctx->rng_state, sizeof ctx->rng_state, (struct random_data*)ctx->rng))
        LHGT_FAIL(LHGT_E_ARG, "initstate_r failed");
    ctx->seeded = true;
    return LHGT_OK;
}

// random_coder (E:1182-1222): per k-mer offset, t = e/3+1 draws of rand()%6, each a row of the
// six permutations of (0,1,2); the first e entries of the concatenated rows are kept.
int lhgt_coder_generate(lhgt_ctx* ctx) {
    if (!ctx || !ctx->seeded) LHGT_FAIL(LHGT_E_STATE, "lhgt_rng_seed must be called before lhgt_coder_generate");
    sampling_join(ctx);
    static const int16_t permu[18] = {0, 1, 2, 0, 2, 1, 1, 2, 0, 1, 0, 2, 2, 0, 1, 2, 1, 0};
    int16_t cc[LHGT_CODER_SLOTS];
    for (int i = 0; i < LHGT_CODER_SLOTS; i++) cc[i] = 100;  // E:1189
    int t = ctx->e / 3 + 1;
    int16_t row[16];
    for (int j = 0; j < ctx->k; j++) {
        for (int z = 0; z < t; z++) {
            int r = rng_next(ctx) % 6;
            for (int w = 0; w < 3; w++) row[3 * z + w] = permu[r * 3 + w];
        }
        for (int i = 0; i < ctx->e; i++) cc[j * ctx->e + i] = row[i];
    }
    return lhgt_coder_set(ctx, cc);
}

// random_coder of count_diff_kmer.cpp (C:216-238): ONE draw of rand() % 6 per k-mer offset, three hashes
int lhgt_coder_generate_count_diff(lhgt_ctx* ctx) {
    if (!ctx || !ctx->seeded) LHGT_FAIL(LHGT_E_STATE, "lhgt_rng_seed must be called before lhgt_coder_generate_count_diff");
    sampling_join(ctx);
    if (ctx->e != 3) LHGT_FAIL(LHGT_E_ARG, "count_diff_kmer has three hashes (C:20)");
    static const int16_t permu[18] = {0, 1, 2, 0, 2, 1, 1, 2, 0, 1, 0, 2, 2, 0, 1, 2, 1, 0};
    int16_t cc[LHGT_CODER_SLOTS];
    for (int i = 0; i < LHGT_CODER_SLOTS; i++) cc[i] = 100;
    for (int j = 0; j < ctx->k; j++) {
        const int r = rng_next(ctx) % 6;
        for (int i = 0; i < 3; i++) cc[j * 3 + i] = permu[r * 3 + i];
    }
    return lhgt_coder_set(ctx, cc);
}

int lhgt_coder_set(lhgt_ctx* ctx, const int16_t* cc) {
    if (!ctx || !cc) LHGT_FAIL(LHGT_E_ARG, "null argument");
    HashParams hp;
    LHGT_TRY(build_hash_params(cc, ctx->k, ctx->e, &hp));
    // the slot list groups the reference's positions by the hashes of the coder it was built under: a packed reference may
    // legitimately be scanned under a new coder (phase B recomputes the hashes), its list may not
    if (ctx->device >= 0 && ctx->sl_state != 0 && (!ctx->have_coder || memcmp(&ctx->hp, &hp, sizeof hp) != 0)) {
        if (hipSetDevice(ctx->device) == hipSuccess && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        slot_list_drop(ctx);
    }
    memcpy(ctx->cc, cc, sizeof ctx->cc);
    ctx->hp = hp;
    ctx->have_coder = true;
    return LHGT_OK;
}

int lhgt_coder_get(lhgt_ctx* ctx, int16_t* cc) {
    if (!ctx || !cc) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (!ctx->have_coder) LHGT_FAIL(LHGT_E_STATE, "no coder set");
    memcpy(cc, ctx->cc, sizeof ctx->cc);
    return LHGT_OK;
}

// get_random (E:1332-1340): 50 M values float((rand() % 100000) / 1000.0).  Every value is
// < 100, so with ratio >= 100 every read passes `r < ratio` (E:1044) and the array is not needed;
// nothing draws from the stream afterwards, so skipping the fill is unobservable -- and so is filling only the entries a run can
// look at: read n looks at entry n % 5*10^7, so a run over fewer reads (lhgt_sampling_reserve) needs only that many.
// The fill runs glibc's TYPE_3 generator (r[i] = r[i-31] + r[i-3], output r[i] >> 1) directly on the private random_r state --
// the same words random_r would produce, without 5*10^7 calls -- and takes value -> float from a table of the 10^5 possible values
// (the reference's `(rand() % 100000) / 1000.0` in double, rounded to float, E:1336).
int lhgt_sampling_reserve(lhgt_ctx* ctx, long n_reads) {
    if (!ctx || n_reads < 0) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    ctx->sampling_reads = n_reads;
    if (ctx->fill_thread && n_reads > 0 && n_reads < LHGT_MAX_RANDOM) ctx->fill_limit.store(n_reads);   // a fill in flight stops there
    return LHGT_OK;
}

}  // extern "C"

namespace lhgt {

// fills random_array[0, limit) from the context's rand() stream; `limit` is re-read every 64 Ki entries (a running fill can be cut short)
static long sampling_fill(lhgt_ctx* ctx, const std::atomic<long>* limit, long fixed_limit) {
    ctx->random_array.assign((size_t)LHGT_MAX_RANDOM, 0.f);
    std::vector<float> lut(100000);
    for (int v = 0; v < 100000; v++) lut[(size_t)v] = (float)(v / 1000.0);
    struct random_data* rd = (struct random_data*)ctx->rng;
    float* out = ctx->random_array.data();
    long i = 0;
    const bool fast = rd->rand_type == 3 && rd->rand_deg == 31 && rd->rand_sep == 3;
    int32_t *f = rd->fptr, *r = rd->rptr, *const st = rd->state, *const end = rd->end_ptr;
    for (;;) {
        const long lim = limit ? limit->load() : fixed_limit;
        if (i >= lim) break;
        const long stop = i + 65536 < lim ? i + 65536 : lim;
        if (fast) {
            for (; i < stop; i++) {
                const uint32_t val = (uint32_t)*f + (uint32_t)*r;
                *f = (int32_t)val;
                out[i] = lut[(val >> 1) % 100000u];
                if (++f >= end) { f = st; ++r; }
                else if (++r >= end) r = st;
            }
        } else {
            for (; i < stop; i++) out[i] = lut[(size_t)(rng_next(ctx) % 100000)];
        }
    }
    if (fast) {
        rd->fptr = f;
        rd->rptr = r;
    }
    return i;
}

// waits for a fill started by lhgt_sampling_begin; every entry point that touches the rand() stream or the array calls it first
void sampling_join(lhgt_ctx* ctx) {
    if (!ctx->fill_thread) return;
    ctx->fill_thread->join();
    delete ctx->fill_thread;
    ctx->fill_thread = nullptr;
    ctx->sampling_filled = ctx->fill_done;
}

}  // namespace lhgt

extern "C" {

// get_random (E:1332-1340) started early: the 5*10^7 draws take 0.3 s of ONE host core and depend on nothing but the seed and the
// coder's draws before them, so they can run next to the line count of the FASTQ files and the reference load.  The fill assumes
// ratio < 100; lhgt_sampling_init(ratio) later joins it (and drops the array when ratio >= 100: nothing observes the draws then).
int lhgt_sampling_begin(lhgt_ctx* ctx) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (!ctx->seeded) LHGT_FAIL(LHGT_E_STATE, "lhgt_rng_seed must be called before lhgt_sampling_begin");
    sampling_join(ctx);
    ctx->fill_limit.store(ctx->sampling_reads > 0 && ctx->sampling_reads < LHGT_MAX_RANDOM ? ctx->sampling_reads : LHGT_MAX_RANDOM);
    ctx->fill_done = 0;
    ctx->fill_thread = new std::thread([ctx] { ctx->fill_done = sampling_fill(ctx, &ctx->fill_limit, 0); });
    return LHGT_OK;
}

int lhgt_sampling_init(lhgt_ctx* ctx, double ratio_percent) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    ctx->ratio = ratio_percent;
    if (ctx->fill_thread) {                      // begun early: cut it short if nothing will look, else let it reach its limit
        if (ratio_percent >= 100.0) ctx->fill_limit.store(0);
        sampling_join(ctx);
        if (ratio_percent >= 100.0) { ctx->random_array.clear(); ctx->random_array.shrink_to_fit(); }
        return LHGT_OK;
    }
    ctx->random_array.clear();
    if (ratio_percent >= 100.0) return LHGT_OK;
    if (!ctx->seeded) LHGT_FAIL(LHGT_E_STATE, "lhgt_rng_seed must be called before lhgt_sampling_init");
    // the array keeps its full length (every consumer indexes it modulo 5*10^7); entries no read of the run can look at stay 0
    const long need = ctx->sampling_reads > 0 && ctx->sampling_reads < LHGT_MAX_RANDOM ? ctx->sampling_reads : LHGT_MAX_RANDOM;
    ctx->sampling_filled = sampling_fill(ctx, nullptr, need);
    return LHGT_OK;
}

int lhgt_sampling_get(lhgt_ctx* ctx, float* out, long n) {
    if (!ctx || !out) LHGT_FAIL(LHGT_E_ARG, "null argument");
    sampling_join(ctx);
    if ((long)ctx->random_array.size() < n) LHGT_FAIL(LHGT_E_STATE, "sampling array not filled (ratio >= 100?)");
    memcpy(out, ctx->random_array.data(), sizeof(float) * (size_t)n);
    return LHGT_OK;
}

}  // extern "C"
