/* Oracle (test infrastructure only): scalar C restatement of the negative-index
 * sampler of /root/reference/cpc/criterion/criterion.py:247-266 as executed by
 * torch's CPU generator (32-bit MT19937, one output per element,
 * value = out % range + low; batchIdx drawn first, then seqIdx).
 *
 * Built by __graft_entry__.build() into oracle/_build/liboracle_mt.so and used
 * only by tests/ and bench.py's cpu_baseline leg.
 */
#include <stdint.h>
#include <stddef.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t mt[MT_N];
    int left;  /* torch bookkeeping: twist when --left == 0 */
    int next;
} oracle_mt;

void oracle_mt_seed(oracle_mt *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < MT_N; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->left = 1;
    g->next = 0;
}

static void twist(oracle_mt *g)
{
    uint32_t *mt = g->mt;
    for (int i = 0; i < MT_N; ++i) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % MT_N] & 0x7fffffffu);
        mt[i] = mt[(i + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}

static uint32_t next_u32(oracle_mt *g)
{
    if (--g->left == 0) {
        twist(g);
        g->left = MT_N;
        g->next = 0;
    }
    uint32_t y = g->mt[g->next++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* ext[i], i = (bb*n_neg + nn)*window + t ; all outputs have n_neg*window*batch entries */
void oracle_negative_indices(oracle_mt *g, int batch, int seq_len, int window, int n_neg,
                             int64_t *batch_idx, int64_t *seq_raw, int64_t *ext)
{
    size_t n = (size_t)n_neg * window * batch;
    for (size_t i = 0; i < n; ++i)
        batch_idx[i] = (int64_t)(next_u32(g) % (uint32_t)batch);
    for (size_t i = 0; i < n; ++i)
        seq_raw[i] = (int64_t)(next_u32(g) % (uint32_t)(seq_len - 1)) + 1;
    for (size_t i = 0; i < n; ++i) {
        int64_t t = (int64_t)(i % (size_t)window);
        ext[i] = (seq_raw[i] + t) % seq_len + batch_idx[i] * seq_len;
    }
}

size_t oracle_mt_sizeof(void) { return sizeof(oracle_mt); }
