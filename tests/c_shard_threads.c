/*
 * tests/c_shard_threads.c -- the form a C host like the reference's skred.c (one process; skred.c:107-119 is its one call site)
 * would give a multi-GPU node: ONE PROCESS, one thread per rank, every thread driving its own skred_shard_t.  INTEGRATION.md
 * promises that skred_shard_* is safe to use that way -- no state shared between shards, the last-error text per thread --;
 * this program rehearses it WITHOUT devices: the render / master steps are host functions (skred_shard_create_custom) that
 * render a closed-form "voice" per voice of the rank's range into host memory, and the collective is a reduce built from two
 * pthread barriers around a sum in rank order on the root -- the place an N-GPU host would call ncclReduce on each rank's
 * stream.  Every rank count must reproduce, bit for bit, what ONE rank renders for the whole bank when the per-rank partial
 * sums are added in rank order (the test builds that truth the same way), through the serial sequence AND the pipelined one,
 * while all threads hammer the library at the same time; a deliberately failing call on one thread must not disturb the
 * error text another thread reads.  Compiled and run by tests/test_c_abi.py (CPU suite); prints "OK" and exits 0.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "skred_amd.h"

#define TOTAL 10007          /* voices: prime, so that the ranks' blocks differ in length */
#define MAXW 8
#define BLOCKS 24
#define CHANNELS 2
static const int lengths[4] = {64, 512, 300, 128};

typedef struct {
  int rank, world, lo, hi;
  unsigned long long frame0;           /* frames rendered so far (this rank's clock) */
  float gain;                          /* the master stage's carried gain (root only) */
} rank_t;

/* ---- the stand-in steps: deterministic, rank-local, order-sensitive enough that a wrong sequence shows */
static float voice_sample(int v, unsigned long long f) {
  const float x = (float)((v * 2654435761u + (unsigned)f * 40503u) & 0xFFFF) / 65536.0f - 0.5f;
  return x * (1.0f / (1.0f + (float)(v & 7)));
}
static int step_render(void *ctx, int num_frames, int interp, float *partial, void *stream) {
  (void)interp; (void)stream;
  rank_t *r = (rank_t *)ctx;
  for (int f = 0; f < num_frames; f++) {
    float l = 0.0f, rr = 0.0f;
    for (int v = r->lo; v < r->hi; v++) {     /* voice order inside the rank: fixed */
      const float s = voice_sample(v, r->frame0 + (unsigned long long)f);
      l += s * 0.25f; rr += s * 0.75f;
    }
    partial[2 * f] = l; partial[2 * f + 1] = rr;
  }
  r->frame0 += (unsigned long long)num_frames;
  return SKRED_OK;
}
static int step_master(void *ctx, const float *sum, int num_frames, int num_channels, float *out, void *stream) {
  (void)stream;
  rank_t *r = (rank_t *)ctx;
  for (int f = 0; f < num_frames; f++) {      /* the serial gain recurrence of synth.c:616-620 */
    r->gain += 0.002f * (0.025f - r->gain);
    out[f * num_channels] = sum[2 * f] * r->gain;
    out[f * num_channels + 1] = sum[2 * f + 1] * r->gain;
  }
  return SKRED_OK;
}

/* ---- the collective: every rank publishes its buffer, the root adds them in rank order between two barriers */
typedef struct {
  pthread_barrier_t in, out;
  float *buf[MAXW];
  int world;
} comm_t;
typedef struct { comm_t *c; int rank; } reduce_ctx_t;
static int step_reduce(void *ctx, float *partial, size_t n, int root, void *stream) {
  (void)stream;
  reduce_ctx_t *rc = (reduce_ctx_t *)ctx;
  comm_t *c = rc->c;
  c->buf[rc->rank] = partial;
  pthread_barrier_wait(&c->in);
  if (rc->rank == root) {
    for (size_t i = 0; i < n; i++) {
      float s = c->buf[0][i];
      for (int k = 1; k < c->world; k++) s += c->buf[k][i];
      partial[i] = s;                          /* (root == 0 in this test: buf[0] is `partial`, read before it is written) */
    }
  }
  pthread_barrier_wait(&c->out);
  return SKRED_OK;
}

/* ---- one job: `world` threads, serial or pipelined sequence; the root's output of every block goes to `got` */
typedef struct {
  comm_t *c; int rank, world, pipelined; float *got; int failed;
} thread_arg_t;

static size_t block_offset(int k) { size_t o = 0; for (int i = 0; i < k; i++) o += (size_t)lengths[i & 3] * CHANNELS; return o; }

static void *rank_main(void *p) {
  thread_arg_t *a = (thread_arg_t *)p;
  rank_t me = { a->rank, a->world, 0, 0, 0ull, 0.0f };
  reduce_ctx_t rctx = { a->c, a->rank };
  skred_shard_ops_t ops;
  memset(&ops, 0, sizeof(ops));
  ops.ctx = &me; ops.render = step_render; ops.master = step_master; ops.reduce_ctx = &rctx; ops.reduce = step_reduce;
  skred_shard_t *s = NULL;
  a->failed = 1;
  if (skred_shard_create_custom(a->rank, a->world, 0, TOTAL, &ops, &s) != SKRED_OK) return NULL;
  if (skred_shard_range(s, &me.lo, &me.hi) != SKRED_OK) return NULL;
  float *partial = malloc(512 * 2 * sizeof(float)), *scratch = malloc(512 * CHANNELS * sizeof(float));
  for (int k = 0; k < BLOCKS; k++) {
    const int F = lengths[k & 3];
    float *out = a->rank == 0 ? a->got + block_offset(k) : scratch;
    int rc = a->pipelined ? skred_shard_render_mix_pipelined(s, F, SKRED_INTERP_TRUNCATE, out, CHANNELS, NULL)
                          : skred_shard_render_mix(s, F, SKRED_INTERP_TRUNCATE, partial, out, CHANNELS, NULL);
    if (rc != SKRED_OK) { fprintf(stderr, "rank %d block %d: %s\n", a->rank, k, skred_amd_last_error()); return NULL; }
    if ((k % 5) == a->rank % 5) {
      /* a call that fails ON THIS THREAD: the text it leaves must be this thread's own, whatever the others do meanwhile */
      if (skred_shard_render_mix(s, -1, SKRED_INTERP_TRUNCATE, partial, out, CHANNELS, NULL) != SKRED_E_BAD_ARG) return NULL;
      if (!strstr(skred_amd_last_error(), "shard_render_mix")) { fprintf(stderr, "rank %d reads another thread's error: %s\n", a->rank, skred_amd_last_error()); return NULL; }
    }
  }
  if (skred_shard_flush(s, NULL) != SKRED_OK) return NULL;
  skred_shard_destroy(s);
  free(partial); free(scratch);
  a->failed = 0;
  return NULL;
}

static int run_job(int world, int pipelined, float *got) {
  comm_t c;
  memset(&c, 0, sizeof(c));
  c.world = world;
  pthread_barrier_init(&c.in, NULL, (unsigned)world);
  pthread_barrier_init(&c.out, NULL, (unsigned)world);
  pthread_t th[MAXW];
  thread_arg_t arg[MAXW];
  for (int r = 0; r < world; r++) {
    arg[r] = (thread_arg_t){ &c, r, world, pipelined, got, 1 };
    if (pthread_create(&th[r], NULL, rank_main, &arg[r]) != 0) return 1;
  }
  int bad = 0;
  for (int r = 0; r < world; r++) { pthread_join(th[r], NULL); bad |= arg[r].failed; }
  pthread_barrier_destroy(&c.in); pthread_barrier_destroy(&c.out);
  return bad;
}

int main(void) {
  const size_t total = block_offset(BLOCKS);
  float *want = malloc(total * sizeof(float)), *got = malloc(total * sizeof(float));
  for (int world = 1; world <= MAXW; world++) {
    /* the truth for this rank count: the same per-rank partial sums, added in rank order, on one thread */
    {
      rank_t rk[MAXW];
      float gain = 0.0f;
      float *acc = malloc(512 * 2 * sizeof(float)), *part = malloc(512 * 2 * sizeof(float));
      for (int r = 0; r < world; r++) { rk[r] = (rank_t){ r, world, 0, 0, 0ull, 0.0f }; skred_shard_partition(TOTAL, world, r, &rk[r].lo, &rk[r].hi); }
      for (int k = 0; k < BLOCKS; k++) {
        const int F = lengths[k & 3];
        for (int r = 0; r < world; r++) {
          step_render(&rk[r], F, 0, part, NULL);
          for (int i = 0; i < 2 * F; i++) acc[i] = r == 0 ? part[i] : acc[i] + part[i];
        }
        rank_t root = { 0, world, 0, 0, 0ull, gain };
        step_master(&root, acc, F, CHANNELS, want + block_offset(k), NULL);
        gain = root.gain;
      }
      free(acc); free(part);
    }
    for (int pipelined = 0; pipelined <= 1; pipelined++) {
      memset(got, 0, total * sizeof(float));
      if (run_job(world, pipelined, got)) { fprintf(stderr, "job world=%d pipelined=%d failed\n", world, pipelined); return 1; }
      if (memcmp(got, want, total * sizeof(float)) != 0) { fprintf(stderr, "world=%d pipelined=%d: output differs from the rank-ordered sum\n", world, pipelined); return 1; }
    }
  }
  printf("OK %d blocks x world 1..%d x (serial, pipelined): one thread per rank, bit-equal to the rank-ordered sum\n", BLOCKS, MAXW);
  return 0;
}
