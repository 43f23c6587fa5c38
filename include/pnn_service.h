/* pnn_service.h -- cross-process batching service for the in-loop PNN calls.
 *
 * Inside one HM encode the PNN calls are serially dependent and arrive one block at a time
 * (TComPrediction.cpp:554-655); the data-parallel axis is ACROSS the many independent encodes an experiment runs
 * (hevc/running.py, comparing_rate_distortion.py:96-123: images x QPs x variants).  One server process owns the GPU
 * context; every encoder process connects over a Unix-domain socket and issues the same single-block call it would
 * issue locally; the server coalesces the requests of one width that are pending at the same time into ONE batched
 * pnn_predict_pel call and routes the results back.  Blocking, one outstanding request per client -- exactly HM's
 * calling pattern.  Same error conventions as pnn_hip.h (0 / negative, nothing throws).
 */
#ifndef PNN_SERVICE_H
#define PNN_SERVICE_H
#include <stdint.h>
#include "pnn_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* What the server calls for a batch: n blocks of one width, inputs stacked as pnn_predict_pel takes them
 * (FC widths: `above` = [n][5w^2] flattened contexts, `left` = NULL).  dst = [n][w][w] Pel values after the HM epilogue,
 * out_f32 = [n][w][w] float predictions as the frozen graph returns them (mean not re-added); either may be NULL when no
 * client of the batch asked for that kind.  Returns 0 or a negative code. */
typedef int (*pnn_service_backend)(void* user, int width, const float* above, const float* left, int n, int32_t* dst,
                                   float* out_f32);

/* Serves `socket_path` until *stop becomes non-zero (checked at least every 50 ms).  max_batch: largest batch handed to
 * the backend; window_us: after the first pending request the server waits up to this long for more before it
 * dispatches (0 = dispatch whatever is pending right now -- the recommended setting: while a batch is on the GPU the next
 * one forms by itself, and with 4 / 16 HM encoders on one server a 100 us window cost 27 / 33 % of the wall time).  stats (optional, 4 longs): requests served, backend
 * calls, largest batch, clients accepted. */
int pnn_service_run_backend(const char* socket_path, pnn_service_backend backend, void* user, int max_batch, int window_us,
                            volatile int* stop, long* stats);
/* The same with backend = pnn_predict_f32_pel on `ctx` (is_fc per width as the loaded models say).  The server never
 * blocks on one client: a client that stalls in the middle of a request, stops reading its replies or sends a malformed
 * header is dropped and the others go on. */
int pnn_service_run(const char* socket_path, pnn_ctx* ctx, int max_batch, int window_us, volatile int* stop, long* stats);
/* The production form: the server creates its own contexts from the model table (selection rule of pnn_create:
 * TComPrediction.cpp:143-178) -- FIVE of them, one per width with that width's model only.  The 4x4, 8x8 and 16x16 widths each
 * have a worker thread and a stream on a HARDWARE QUEUE of their own (pnn_streams_on_distinct_queues: the runtime has four, and two
 * busy streams on one queue wait for each other's whole calls); 32x32 and 64x64 share the fourth queue and a thread.  Passes of
 * different widths overlap on the GPU and with the socket work; while a worker is busy the requests for its width accumulate:
 * the next batch forms by itself.  ($PNN_SERVICE_QUEUES=0: every context on the stream it created, one thread per width.)  Socket work (accept, receive, reply) is spread
 * over 4 I/O threads, each owning its share of the connections ($PNN_SERVICE_IO_THREADS: 1 ... 8): one socket thread topped
 * out near 120 k requests/s, the ceiling of a server behind 24 or 100 HM encoders alike.  A request
 * whose shape (n_above, n_left) does not fit the kind of model loaded for its width is answered with PNN_E_ARG / PNN_E_MODEL
 * at once and never reaches the backend. */
int pnn_service_run_table(const char* socket_path, const char* model_table_path, int use_pair, float mean, int device, int max_batch,
                          int window_us, volatile int* stop, long* stats);

/* Client side: what an encoder process links instead of owning a GPU context. */
typedef struct pnn_client pnn_client;
int pnn_client_connect(pnn_client** out, const char* socket_path);
/* == pnn_predict_pel(ctx, width, above, left, 1, dst, dst_stride) executed by the server (left = NULL for FC widths,
 * where `above` is the [5w^2] flattened context). */
int pnn_client_predict_pel(pnn_client* c, int width, const float* above, const float* left, int32_t* dst, int dst_stride);
/* == Session::Run on the server: the float prediction [w][w] (what the TensorFlow look-alike of pnn_tf_compat.h binds when
 * PNN_SERVICE_SOCKET is set, so that an UNMODIFIED HM process is served by the batching service). */
int pnn_client_predict_f32(pnn_client* c, int width, const float* above, const float* left, float* out);
/* The arithmetic tag (pnn_arithmetic_tag, pnn_hip.h) of the server context that answers requests of `width`: everything that decides
 * the last float bits of its predictions.  An encoder behind the service and the decoder of its bitstream predict identically iff
 * their tags are equal -- ask once at start-up and compare (the TensorFlow look-alike does, against $PNN_EXPECT_TAG).
 * A server with a generic backend answers $PNN_SERVICE_TAG of its process, or "backend:unspecified". */
int pnn_client_arithmetic_tag(pnn_client* c, int width, char* out, size_t bytes);
/* Repeated requests (same width, same input bytes -- HM's RD search re-asks) are answered from a
 * client-side cache of $PNN_CACHE_MB MiB (default 64, 0 = off) without a round trip; hits / misses so far. */
int pnn_client_cache_stats(const pnn_client* c, long* hits, long* misses);
void pnn_client_close(pnn_client* c);

#ifdef __cplusplus
}
#endif
#endif /* PNN_SERVICE_H */
