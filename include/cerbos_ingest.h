/*
 * cerbos_ingest.h - C ABI of libcerbos_ingest.so: host-side ingest for libcerbos_hip.so.
 *
 * Turns serialized enginev1.CheckInput messages (api/public/cerbos/engine/v1/engine.proto:130-160,
 * the element type of evaluator.Evaluator.Check's input slice, internal/evaluator/evaluator.go:16-19)
 * into the SoA `cbh_batch` that cbh_check_batch / cbh_batch_upload consume, so that a Go caller hands
 * over the bytes it already has (one cgo call per batch, no per-field marshalling): SURVEY.md §8(f)-1.
 * It restates, in C++, what the reference does per request before the rule table is consulted:
 *   checkInputToRequest            internal/ruletable/check.go:536-554   (request view of a CheckInput)
 *   evaluator.Scope / PolicyVersion internal/evaluator/evaluator.go:108-122 (defaults)
 *   namer.SanitizedResource        internal/namer/namer.go:213-218
 *   namer.ScopeValue / ScopeParents internal/namer/namer.go:77-87, 276-278
 * plus the interning, attribute-column extraction and routing sort of cerbos_amd/flatten.py, whose
 * output it reproduces array for array (tests/test_ingest.py).
 *
 * Pure host code (no HIP): plain C types, caller-owned input buffers are never retained, 0 = OK,
 * < 0 = error with text in cbi_last_error() (thread-local).
 */
#ifndef CERBOS_INGEST_H
#define CERBOS_INGEST_H

#include <stddef.h>
#include <stdint.h>

#include "cerbos_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cbi_table cbi_table; /* host dictionaries of one lowered table image */
typedef struct cbi_batch cbi_batch; /* owns the arrays of one flattened batch */

const char* cbi_last_error(void);

/* Builds the host dictionaries (string pool, scope index, attribute-column paths) from the same
 * table image that cbh_table_load takes.  The image is copied.  A cbi_table is immutable afterwards: any
 * number of threads may flatten against it concurrently. */
int cbi_table_open(const void* blob, size_t len, cbi_table** out);
void cbi_table_close(cbi_table* t);

/*
 * Flattens `n` serialized CheckInput messages: message i = bytes[offsets[i] .. offsets[i+1]).
 * default_version / default_scope = EvalParams.DefaultPolicyVersion / DefaultScope.
 * sort != 0 orders the device requests by route (kind, resource version, resource scope, role list)
 * as the kernels like it; results are mapped back with cbi_batch_tuple_perm.
 * A CheckInput with more than CBH_MAX_ACTIONS_PER_REQUEST actions becomes several device requests.
 */
int cbi_flatten_pb(const cbi_table* t, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                   const char* default_version, const char* default_scope, int sort, cbi_batch** out);
void cbi_batch_free(cbi_batch* b);

/* The flattened batch; valid until cbi_batch_free. */
const cbh_batch* cbi_batch_view(const cbi_batch* b);
/* Device tuple j holds the input-order tuple tuple_perm[j] (input order = inputs in order, each one's
 * actions in order).  n_tuples entries. */
const uint64_t* cbi_batch_tuple_perm(const cbi_batch* b);
/* Index of the CheckInput each device request came from.  n_requests entries. */
const uint32_t* cbi_batch_request_input(const cbi_batch* b);

#ifdef __cplusplus
}
#endif
#endif /* CERBOS_INGEST_H */
