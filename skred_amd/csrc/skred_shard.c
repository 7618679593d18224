/*
 * skred_shard.c -- voices sharded over the GPUs of one node, from C (include/skred_amd.h: skred_shard_*).
 *
 * The render loop shards naturally: voices are independent unless they name each other as
 * modulators, so rank r of `world` renders the contiguous block [lo, hi) of the bank into a PRE-master partial
 * mix float[F][2]; the only exchange step of the path is the sum of those partials on the root -- ONE reduce of
 * 8*F bytes per block (RCCL over xGMI; latency-bound) -- after which the root runs the serial master-volume
 * stage (synth.c:616-624) once.  One process per GPU, as the rest of this build assumes.
 *
 * What lives here is the host side a C program such as skred itself (skred.c:107-116 is its audio callback) would
 * otherwise have to write: the partition rule, the check that no modulation crosses a cut, the per-block
 * sequence render -> reduce -> master, and the RCCL call.  The three steps are function pointers (skred_shard_ops_t):
 * skred_shard_create() fills them with the bank-mode entry points and, after skred_shard_init_rccl(), ncclReduce;
 * tests fill them with the oracle and a gloo reduce to rehearse the very same sequencing on CPUs
 * (tests/test_sharded_gloo.py through skred_amd/sharded.py, a ctypes image of this file).
 *
 * RCCL is reached through dlopen("librccl.so.1"): the library has no link-time dependency on it, and a process that
 * already carries an RCCL (PyTorch does) gets that very copy.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "skred_bank_priv.h"
#include "skred_amd_fxpt.h"

struct skred_shard {
  int rank, world, root, total, lo, hi;
  int always_reduce;                 /* run the collective even with one rank (rehearsal on a one-GPU box) */
  skred_shard_ops_t ops;
  /* bank-backed flavour */
  skred_bank_t *bank;
  skred_fxbank_t *fxbank;            /* ... of the fixed-point path (skred_fxshard_create): int64 partials, an exact sum */
  int elem_bytes;                    /* of one partial-mix element: 4 (float) or 8 (int64) */
  int device;
  float *d_partial;                  /* [F][2] scratch when the caller passes none */
  size_t partial_cap;
  /* pipelined form (skred_shard_render_mix_pipelined): two partial buffers, the collective's own stream, event chain */
  float *pp_buf[2];                  /* device memory (bank-backed steps) or host memory (custom steps) */
  size_t pp_cap;
  unsigned pp_k;                     /* blocks issued */
  int pp_in_flight;                  /* a pipelined block was issued since the last serial one */
  hipStream_t pp_comm;
  hipEvent_t pp_rendered[2], pp_done[2];
  /* library-owned RCCL communicator */
  void *rccl_lib;
  void *comm;
  int (*nccl_reduce)(const void *, void *, size_t, int, int, int, void *, void *);
  int (*nccl_comm_destroy)(void *);
  const char *(*nccl_error_string)(int);
};

/* ------------------------------------------------------------------ partition, legality of a cut */

int skred_shard_partition(int total_voices, int world, int rank, int *lo, int *hi) {
  if (total_voices < 0 || world <= 0 || rank < 0 || rank >= world || !lo || !hi) return fail(SKRED_E_BAD_ARG, "shard_partition: bad arguments");
  *lo = (int)((long long)total_voices * rank / world);            /* contiguous blocks that differ by at most one voice */
  *hi = (int)((long long)total_voices * (rank + 1) / world);
  return SKRED_OK;
}

/* 1 when no voice of [lo,hi) is modulated by a voice outside it (FM / AM / pan; the CZ source only with CZ on,
 * synth.c:262): only then is the block an independent bank.  0 otherwise. */
int skred_shard_cut_ok(const skred_voice_bank_t *h, int lo, int hi) {
  if (!h || lo < 0 || hi > h->n_voices || lo > hi) return 0;
  for (int v = lo; v < hi; v++) {
    const int src[4] = { h->voice_freq_mod_osc[v], h->voice_amp_mod_osc[v], h->voice_pan_mod_osc[v],
                         h->voice_cz_mode[v] ? h->voice_cz_mod_osc[v] : -1 };
    for (int k = 0; k < 4; k++)
      if (src[k] >= 0 && (src[k] < lo || src[k] >= hi)) return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------ the three steps on a device bank */

static int bank_render(void *ctx, int num_frames, int interp, float *partial, void *stream) {
  return skred_bank_render(((skred_shard_t *)ctx)->bank, num_frames, interp, partial, NULL, stream);
}

static int bank_master(void *ctx, const float *sum, int num_frames, int num_channels, float *out, void *stream) {
  return skred_bank_master(((skred_shard_t *)ctx)->bank, sum, num_frames, num_channels, out, stream);
}

static int rccl_reduce(void *ctx, float *partial, size_t n_floats, int root, void *stream) {
  skred_shard_t *s = (skred_shard_t *)ctx;
  if (!s->comm) return fail(SKRED_E_BAD_ARG, "shard: %d ranks but no collective (skred_shard_init_rccl / skred_shard_set_ops)", s->world);
  /* in place on the root; ncclFloat32 = 7, ncclInt64 = 4, ncclSum = 0 (rccl.h) */
  const int rc = s->nccl_reduce(partial, partial, n_floats, s->elem_bytes == 8 ? 4 : 7, 0, root, s->comm, stream);
  if (rc != 0) return fail(SKRED_E_NO_DEVICE, "ncclReduce -> %s", s->nccl_error_string ? s->nccl_error_string(rc) : "error");
  return SKRED_OK;
}

/* ------------------------------------------------------------------ lifecycle */

static int shard_new(int rank, int world, int root, int total_voices, skred_shard_t **out) {
  if (!out) return fail(SKRED_E_BAD_ARG, "shard_create: bad arguments");
  *out = NULL;
  if (world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world || total_voices < world)
    return fail(SKRED_E_BAD_ARG, "shard_create: rank %d of %d, root %d, %d voices", rank, world, root, total_voices);
  skred_shard_t *s = (skred_shard_t *)calloc(1, sizeof(*s));
  if (!s) return fail(SKRED_E_NO_MEM, "calloc");
  s->rank = rank; s->world = world; s->root = root; s->total = total_voices;
  s->elem_bytes = (int)sizeof(float);
  (void)skred_shard_partition(total_voices, world, rank, &s->lo, &s->hi);
  *out = s;
  return SKRED_OK;
}

int skred_shard_create(int device, int rank, int world, int root, int total_voices, skred_shard_t **out) {
  int rc = shard_new(rank, world, root, total_voices, out);
  if (rc) return rc;
  skred_shard_t *s = *out;
  s->device = device;
  rc = skred_bank_create(device, s->hi - s->lo, &s->bank);
  if (rc) { free(s); *out = NULL; return rc; }
  s->ops.ctx = s;
  s->ops.render = bank_render;
  s->ops.master = bank_master;
  s->ops.reduce_ctx = s;
  s->ops.reduce = rccl_reduce;       /* fails loudly until a communicator exists; never called with one rank */
  return SKRED_OK;
}

/* ------------------------------------------------------------------ the fixed-point path, sharded the same way
 *
 * Every rank renders its block of an skred_fxpt_bank_t into an int64 pre-master sum; the reduce is ncclSum on int64 -- an EXACT
 * sum, so the sharded render equals the unsharded one bit for bit whatever the number of ranks --; the root applies the integer
 * master stage (include/skred_amd_fxpt.h).  Same skred_shard_t, same skred_shard_render_mix / _init_rccl / _set_ops:
 * `partial` and `out` are int64[num_frames][2] there (passed through the float pointers of the float path's signature). */
static int fx_render_step(void *ctx, int num_frames, int interp, float *partial, void *stream) {
  return skred_fxbank_render(((skred_shard_t *)ctx)->fxbank, num_frames, interp, (int64_t *)partial, NULL, stream);
}
static int fx_master_step(void *ctx, const float *sum, int num_frames, int num_channels, float *out, void *stream) {
  (void)num_channels;
  return skred_fxbank_master(((skred_shard_t *)ctx)->fxbank, (const int64_t *)sum, num_frames, (int64_t *)out, stream);
}

int skred_fxshard_create(int device, int rank, int world, int root, int total_voices, skred_shard_t **out) {
  int rc = shard_new(rank, world, root, total_voices, out);
  if (rc) return rc;
  skred_shard_t *s = *out;
  s->device = device;
  s->elem_bytes = (int)sizeof(int64_t);
  rc = skred_fxbank_create(device, s->hi - s->lo, &s->fxbank);
  if (rc) { free(s); *out = NULL; return rc; }
  s->ops.ctx = s;
  s->ops.render = fx_render_step;
  s->ops.master = fx_master_step;
  s->ops.reduce_ctx = s;
  s->ops.reduce = rccl_reduce;
  return SKRED_OK;
}

skred_fxbank_t *skred_fxshard_bank(skred_shard_t *s) { return s ? s->fxbank : NULL; }

/* this rank's block of the WHOLE fixed-point bank (voices of that path are independent: every cut is legal) */
int skred_fxshard_upload(skred_shard_t *s, const skred_fxpt_bank_t *whole) {
  if (!s || !whole || !s->fxbank) return fail(SKRED_E_BAD_ARG, "fxshard_upload: bad arguments");
  if (whole->n_voices != s->total) return fail(SKRED_E_RANGE, "fxshard_upload: the bank has %d voices, the shard was made for %d", whole->n_voices, s->total);
  return skred_fxbank_upload(s->fxbank, whole, s->lo, 0, s->hi - s->lo);
}

int skred_shard_create_custom(int rank, int world, int root, int total_voices, const skred_shard_ops_t *ops, skred_shard_t **out) {
  if (!ops || !ops->render || !ops->master || (world > 1 && !ops->reduce)) return fail(SKRED_E_BAD_ARG, "shard_create_custom: missing step");
  const int rc = shard_new(rank, world, root, total_voices, out);
  if (rc) return rc;
  (*out)->ops = *ops;
  (*out)->device = -1;
  return SKRED_OK;
}

void skred_shard_destroy(skred_shard_t *s) {
  if (!s) return;
  if (s->comm && s->nccl_comm_destroy) (void)s->nccl_comm_destroy(s->comm);
  if (s->d_partial) { (void)hipSetDevice(s->device); (void)hipFree(s->d_partial); }
  for (int i = 0; i < 2; i++) {
    if (s->pp_buf[i]) { if (s->bank) (void)hipFree(s->pp_buf[i]); else free(s->pp_buf[i]); }
    if (s->pp_rendered[i]) (void)hipEventDestroy(s->pp_rendered[i]);
    if (s->pp_done[i]) (void)hipEventDestroy(s->pp_done[i]);
  }
  if (s->pp_comm) (void)hipStreamDestroy(s->pp_comm);
  if (s->bank) skred_bank_destroy(s->bank);
  if (s->fxbank) skred_fxbank_destroy(s->fxbank);
  /* the RCCL handle stays open: the process may hold other communicators on it */
  free(s);
}

skred_bank_t *skred_shard_bank(skred_shard_t *s) { return s ? s->bank : NULL; }

int skred_shard_range(const skred_shard_t *s, int *lo, int *hi) {
  if (!s) return fail(SKRED_E_BAD_ARG, "shard_range");
  if (lo) *lo = s->lo;
  if (hi) *hi = s->hi;
  return SKRED_OK;
}

int skred_shard_set_ops(skred_shard_t *s, const skred_shard_ops_t *ops, int always_reduce) {
  if (!s) return fail(SKRED_E_BAD_ARG, "shard_set_ops");
  if (ops) {
    if (ops->render || ops->master) {                  /* the compute steps come as a pair, with their context */
      if (!ops->render || !ops->master) return fail(SKRED_E_BAD_ARG, "shard_set_ops: render and master come together");
      s->ops.render = ops->render; s->ops.master = ops->master; s->ops.ctx = ops->ctx;
    }
    if (ops->reduce) { s->ops.reduce = ops->reduce; s->ops.reduce_ctx = ops->reduce_ctx; }
  }
  s->always_reduce = always_reduce != 0;
  return SKRED_OK;
}

/* this rank's block of the WHOLE bank (the same host view on every rank) goes to its device */
int skred_shard_upload(skred_shard_t *s, const skred_voice_bank_t *whole) {
  if (!s || !whole || !s->bank) return fail(SKRED_E_BAD_ARG, "shard_upload: bad arguments");
  if (whole->n_voices != s->total) return fail(SKRED_E_RANGE, "shard_upload: the bank has %d voices, the shard was made for %d", whole->n_voices, s->total);
  if (!skred_shard_cut_ok(whole, s->lo, s->hi))
    return fail(SKRED_E_UNSUPPORTED, "voices [%d,%d) are modulated across the cut: keep modulator and carrier on one GPU (SURVEY 8e)", s->lo, s->hi);
  /* modulator indices are bank-wide: inside the block they keep their distance to the carrier (sk_pack_voice) */
  return skred_bank_upload(s->bank, whole, s->lo, 0, s->hi - s->lo);
}

/* ------------------------------------------------------------------ RCCL */

#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } nccl_id_t;

static void *rccl_open(void) {
  static const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", NULL };
  for (int i = 0; names[i]; i++) {
    void *h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (h) return h;
  }
  return NULL;
}

int skred_shard_rccl_unique_id(void *out128) {
  if (!out128) return fail(SKRED_E_BAD_ARG, "rccl_unique_id");
  void *lib = rccl_open();
  if (!lib) return fail(SKRED_E_NO_DEVICE, "librccl.so not found: %s", dlerror());
  int (*get_id)(nccl_id_t *) = (int (*)(nccl_id_t *))dlsym(lib, "ncclGetUniqueId");
  if (!get_id) return fail(SKRED_E_NO_DEVICE, "ncclGetUniqueId missing");
  nccl_id_t id;
  const int rc = get_id(&id);
  if (rc != 0) return fail(SKRED_E_NO_DEVICE, "ncclGetUniqueId -> %d", rc);
  memcpy(out128, &id, NCCL_UNIQUE_ID_BYTES);
  return SKRED_OK;
}

/* every rank calls this with the id rank 0 obtained from skred_shard_rccl_unique_id (how the 128 bytes travel is the
 * host program's business: a file, a socket, MPI, torch.distributed) */
int skred_shard_init_rccl(skred_shard_t *s, const void *unique_id128) {
  if (!s || !unique_id128 || (!s->bank && !s->fxbank)) return fail(SKRED_E_BAD_ARG, "shard_init_rccl: bad arguments");
  if (s->comm) return SKRED_OK;
  s->rccl_lib = rccl_open();
  if (!s->rccl_lib) return fail(SKRED_E_NO_DEVICE, "librccl.so not found: %s", dlerror());
  int (*init_rank)(void **, int, nccl_id_t, int) = (int (*)(void **, int, nccl_id_t, int))dlsym(s->rccl_lib, "ncclCommInitRank");
  s->nccl_reduce = (int (*)(const void *, void *, size_t, int, int, int, void *, void *))dlsym(s->rccl_lib, "ncclReduce");
  s->nccl_comm_destroy = (int (*)(void *))dlsym(s->rccl_lib, "ncclCommDestroy");
  s->nccl_error_string = (const char *(*)(int))dlsym(s->rccl_lib, "ncclGetErrorString");
  if (!init_rank || !s->nccl_reduce) return fail(SKRED_E_NO_DEVICE, "RCCL entry points missing");
  HIP_TRY(hipSetDevice(s->device));
  nccl_id_t id;
  memcpy(&id, unique_id128, NCCL_UNIQUE_ID_BYTES);
  const int rc = init_rank(&s->comm, s->world, id, s->rank);
  if (rc != 0) { s->comm = NULL; return fail(SKRED_E_NO_DEVICE, "ncclCommInitRank -> %s", s->nccl_error_string ? s->nccl_error_string(rc) : "error"); }
  return SKRED_OK;
}

/* ------------------------------------------------------------------ one block */

/* render -> (sum over ranks on the root) -> master on the root.  `partial`: float[num_frames][2] scratch in the memory
 * the steps work on (device memory for the bank-backed steps; NULL = the shard's own).  `out` is written on the root
 * only.  Asynchronous on `stream` for the bank-backed steps. */
int skred_shard_render_mix(skred_shard_t *s, int num_frames, int interp, float *partial, float *out, int num_channels, void *stream) {
  if (!s || num_frames <= 0 || num_channels < 2) return fail(SKRED_E_BAD_ARG, "shard_render_mix: bad arguments");
  if (s->rank == s->root && !out) return fail(SKRED_E_BAD_ARG, "shard_render_mix: the root needs an output buffer");
  if (!partial) {
    if (!s->bank && !s->fxbank) return fail(SKRED_E_BAD_ARG, "shard_render_mix: custom steps need a partial buffer");
    const size_t need = (size_t)num_frames * 2;
    if (need > s->partial_cap) {
      HIP_TRY(hipSetDevice(s->device));
      if (s->d_partial) { (void)hipFree(s->d_partial); s->d_partial = NULL; s->partial_cap = 0; }
      HIP_TRY(hipMalloc((void **)&s->d_partial, need * (size_t)s->elem_bytes));
      s->partial_cap = need;
    }
    partial = s->d_partial;
  }
  if (s->pp_comm && s->pp_in_flight) {        /* pipelined calls came before: their last blocks may still be on the shard's stream, */
    HIP_TRY(hipSetDevice(s->device));         /* reading gain rows and state this block's render is about to write */
    HIP_TRY(hipEventSynchronize(s->pp_done[0]));
    HIP_TRY(hipEventSynchronize(s->pp_done[1]));
    s->pp_in_flight = 0;
  }
  int rc = s->ops.render(s->ops.ctx, num_frames, interp, partial, stream);
  if (rc) return rc;
  if (s->world > 1 || s->always_reduce) {
    if (!s->ops.reduce) return fail(SKRED_E_BAD_ARG, "shard_render_mix: a reduce step is required (always_reduce) but none was given");
    rc = s->ops.reduce(s->ops.reduce_ctx, partial, (size_t)num_frames * 2, s->root, stream);
    if (rc) return rc;
  }
  if (s->rank == s->root) rc = s->ops.master(s->ops.ctx, partial, num_frames, num_channels, out, stream);
  return rc;
}

/* ------------------------------------------------------------------ one block, pipelined */

/* The same three steps with the collective of block k overlapped with the render of block k + 1: the render runs on `stream`
 * into one of two partial buffers; reduce and master stage follow on the shard's own stream behind an event; `stream` only
 * waits for them one call later.  Throughput is bounded by max(render, reduce + master) instead of their sum -- the block
 * of a strong-scaling shard is short (2^17 voices: ~56 us) and the 8-rank reduce of 4 KiB is pure latency, so the sum is what
 * kept 8 GPUs at ~3.3x one ("Mid-size banks").
 *
 * Contract (host-paced, see below): `out` of call k is complete -- for any stream, and for the host -- once call k + 2 has
 * returned, or on `stream` after skred_shard_flush(shard, stream).  Call k + 1 returning is NOT enough: nothing on `stream`
 * waits for the shard's own stream, the host only waits (inside call k) for block k - 2.  Consecutive calls must therefore
 * alternate between (at least) two output buffers.  Same samples as skred_shard_render_mix, bit for bit
 * (tests/c_shard_smoke.c).  With custom steps (host memory, synchronous) the calls simply run in order: the CPU rehearsal
 * exercises the buffer rotation, not the overlap. */
static int pp_setup(skred_shard_t *s, size_t need) {
  if (s->bank) {
    /* the collective's stream and the four events, each created once (a call that failed half-way finds the rest on the next one).
     * High priority: the collective and the few workgroups of the master stage must not queue behind the next block's render --
     * and HIP keeps streams of different priorities on different hardware queues, which a second normal-priority stream is not
     * promised (sharing the caller's hardware queue, block k's master stage waited for block k + 1's render: 156 instead of
     * 87 us per block on a 2^18-voice shard inside bench.py, where several streams exist) */
    HIP_TRY(hipSetDevice(s->device));
    if (!s->pp_comm) HIP_TRY(hipStreamCreateWithPriority(&s->pp_comm, hipStreamNonBlocking, -1));
    for (int i = 0; i < 2; i++) {
      if (!s->pp_rendered[i]) HIP_TRY(hipEventCreateWithFlags(&s->pp_rendered[i], hipEventDisableTiming));
      if (!s->pp_done[i]) HIP_TRY(hipEventCreateWithFlags(&s->pp_done[i], hipEventDisableTiming));
    }
  }
  if (need <= s->pp_cap) return SKRED_OK;
  s->pp_cap = 0;                       /* nothing is usable until BOTH buffers exist at the new size */
  if (s->bank) HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < 2; i++) {
    if (s->bank) {
      if (s->pp_buf[i]) { (void)hipFree(s->pp_buf[i]); s->pp_buf[i] = NULL; }
      HIP_TRY(hipMalloc((void **)&s->pp_buf[i], need * sizeof(float)));
    } else {
      free(s->pp_buf[i]);
      s->pp_buf[i] = (float *)malloc(need * sizeof(float));
      if (!s->pp_buf[i]) return fail(SKRED_E_NO_MEM, "pipelined partial buffers");
    }
  }
  s->pp_cap = need;
  return SKRED_OK;
}

int skred_shard_render_mix_pipelined(skred_shard_t *s, int num_frames, int interp, float *out, int num_channels, void *stream) {
  if (!s || num_frames <= 0 || num_channels < 2) return fail(SKRED_E_BAD_ARG, "shard_render_mix_pipelined: bad arguments");
  if (s->fxbank) return fail(SKRED_E_UNSUPPORTED, "shard_render_mix_pipelined: the fixed-point shard has the serial form only");
  if (s->rank == s->root && !out) return fail(SKRED_E_BAD_ARG, "shard_render_mix_pipelined: the root needs an output buffer");
  int rc = pp_setup(s, (size_t)num_frames * 2);
  if (rc) return rc;
  const int p = (int)(s->pp_k & 1u);
  const int collective = s->world > 1 || s->always_reduce;
  if (collective && !s->ops.reduce) return fail(SKRED_E_BAD_ARG, "shard_render_mix_pipelined: a reduce step is required but none was given");
  const int device_steps = s->bank && s->ops.render == bank_render && s->ops.master == bank_master;
  if (!device_steps) {
    /* custom steps: host memory, synchronous -- in order, through the rotating buffers */
    if ((rc = s->ops.render(s->ops.ctx, num_frames, interp, s->pp_buf[p], stream))) return rc;
    if (collective && (rc = s->ops.reduce(s->ops.reduce_ctx, s->pp_buf[p], (size_t)num_frames * 2, s->root, stream))) return rc;
    if (s->rank == s->root) rc = s->ops.master(s->ops.ctx, s->pp_buf[p], num_frames, num_channels, out, stream);
    s->pp_k++;
    return rc;
  }
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t main_s = (hipStream_t)stream;
  /* Host-paced: before block k is issued the HOST waits until block k - 2 has left the collective's stream -- its partial
   * sum, its gain row and the caller's output buffer of that parity are free again, and that output is complete.  The
   * caller's stream itself never waits for the collective's stream: an event wait in front of every render kernel cost it
   * ~6 us of gap per block (2^17-voice shard: 65.2 -> 59.6 us per block), and a host two blocks ahead of the device
   * keeps it busy. */
  if (s->pp_k >= 2) HIP_TRY(hipEventSynchronize(s->pp_done[p]));
  if ((rc = sk_bank_render_sum_pp(s->bank, num_frames, interp, s->pp_buf[p], p, main_s))) return rc;
  HIP_TRY(hipEventRecord(s->pp_rendered[p], main_s));
  HIP_TRY(hipStreamWaitEvent(s->pp_comm, s->pp_rendered[p], 0));
  if (collective && (rc = s->ops.reduce(s->ops.reduce_ctx, s->pp_buf[p], (size_t)num_frames * 2, s->root, s->pp_comm))) return rc;
  if (s->rank == s->root && (rc = sk_bank_master_pp(s->bank, s->pp_buf[p], num_frames, num_channels, out, p, s->pp_comm))) return rc;
  HIP_TRY(hipEventRecord(s->pp_done[p], s->pp_comm));
  s->pp_k++;
  s->pp_in_flight = 1;
  return SKRED_OK;
}

/* `stream` waits for everything the pipelined calls have issued: every block's output is complete on it afterwards (the
 * collective's stream runs its blocks in order, so the last block's event covers them all) */
int skred_shard_flush(skred_shard_t *s, void *stream) {
  if (!s) return fail(SKRED_E_BAD_ARG, "shard_flush");
  if (s->pp_k >= 1 && s->pp_comm) {
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, s->pp_done[(s->pp_k - 1) & 1u], 0));
  }
  return SKRED_OK;
}
