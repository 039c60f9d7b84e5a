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
/* Same result, bit for bit, computed on up to n_threads threads inside the call: slices of the input are
 * flattened concurrently and merged (batch-local string ids, heap offsets and role / tuple offsets rebased so
 * that the batch equals the single-pass one).  n_threads <= 1, or fewer than ~1k messages per thread: single pass. */
int cbi_flatten_pb_mt(const cbi_table* t, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                      const char* default_version, const char* default_scope, int sort, int n_threads, cbi_batch** out);
/* The same with the CALL's globals (evaluator.EvalParams.Globals, internal/evaluator/evaluator.go:52-57, 98-106) as a serialized
 * google.protobuf.Struct: a table lowered with per-call globals reads `G.x` from them (columns of root 4); any other table ignores
 * them (its globals are constants of the image). */
int cbi_flatten_pb_g(const cbi_table* t, const uint8_t* bytes, const uint64_t* offsets, uint32_t n, const char* default_version,
                     const char* default_scope, const uint8_t* globals_pb, uint64_t globals_len, int sort, int n_threads, cbi_batch** out);
/*
 * The same for ONE serialized cerbos.request.v1.CheckResourcesRequest (request.proto:222-273): every resource entry
 * becomes the CheckInput that svc.CheckResources builds from it (cerbos_svc.go:274-287: the request's principal and
 * request id, the entry's resource and actions) without those messages ever being built.  `aux_data` / `aux_len`:
 * the serialized cerbos.engine.v1.AuxData the server derived from the request's JWT (verification is the caller's
 * business), or NULL.  Input index (cbi_batch_request_input) = index of the resource entry.
 */
int cbi_flatten_request_pb(const cbi_table* t, const uint8_t* request, uint64_t request_len, const uint8_t* aux_data,
                           uint64_t aux_len, const char* default_version, const char* default_scope, int sort,
                           int n_threads, cbi_batch** out);
void cbi_batch_free(cbi_batch* b);

/* The flattened batch; valid until cbi_batch_free. */
const cbh_batch* cbi_batch_view(const cbi_batch* b);
/* Device tuple j holds the input-order tuple tuple_perm[j] (input order = inputs in order, each one's
 * actions in order).  n_tuples entries. */
const uint64_t* cbi_batch_tuple_perm(const cbi_batch* b);
/* Index of the CheckInput each device request came from.  n_requests entries. */
const uint32_t* cbi_batch_request_input(const cbi_batch* b);

/*
 * Response assembly: one serialized enginev1.CheckOutput (engine.proto:152-166) per
 * CheckInput, in input order, from the device results of the batch that cbi_flatten_pb made of the same
 * inputs - what checkWithAuditTrail builds from the per-action results (internal/ruletable/check.go:58-95):
 * request_id, resource_id, actions{effect, policy, scope} with duplicates of an action folded "DENY sticky"
 * (check.go:513-530), effective_derived_roles.  `res` is in DEVICE order exactly as cbh_check_batch /
 * cbh_result_download filled it (policy / scope / status / edr_mask may be NULL: those fields are then left out).
 * Not produced here: validation_errors; outputs and evaluation_errors come from the trace pass (cbi_trace_pb below) for
 * the inputs that can have any (cbi_table_trace_scope).
 */
typedef struct cbi_outputs cbi_outputs;
#define CBI_OUT_UNSUPPORTED 1u /* device hit an operation outside its subset: output invalid, caller's engine must run this input */
#define CBI_OUT_CEL_ERROR 2u   /* a CEL runtime error was absorbed on the decision path of at least one action */
#define CBI_OUT_WANTS_TRACE 16u /* an action is marked CBH_ST_WANTS_TRACE: the trace pass has (or may have) outputs / errors for this input */

int cbi_assemble_pb(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint8_t* bytes,
                    const uint64_t* offsets, uint32_t n, const char* default_version, cbi_outputs** out);
/* Same bytes, assembled on up to n_threads threads inside the call (contiguous ranges of inputs). */
int cbi_assemble_pb_mt(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint8_t* bytes,
                       const uint64_t* offsets, uint32_t n, const char* default_version, int n_threads, cbi_outputs** out);
/* The same CheckOutputs for a batch the DEVICE flattened (cerbos_hip.h cbh_wire_flatten: the messages never went through
 * cbi_flatten_pb).  `res` as cbh_result_download filled it for that batch (tuples in input order); in_span / act_span / act_off
 * as cbh_wire_spans_download returned them: the strings an output repeats are read where the device found them. */
int cbi_assemble_wire_pb(const cbi_table* t, const cbh_result* res, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                         uint32_t n_tuples, const uint32_t* in_span, const uint32_t* act_span, const uint32_t* act_off,
                         const char* default_version, int n_threads, cbi_outputs** out);
/* One serialized cerbos.response.v1.CheckResourcesResponse (response.proto:187-300) for the request that
 * cbi_flatten_request_pb flattened: request_id, and per resource entry the resource, actions -> effect and - when
 * the request has include_meta - meta {actions -> matched policy / scope, effective_derived_roles}
 * (cerbos_svc.go:297-343).  cbi_outputs_offsets = {0, length}; cbi_outputs_flags: one byte per resource entry. */
int cbi_assemble_response_pb(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint8_t* request,
                             uint64_t request_len, const char* default_version, cbi_outputs** out);
/* The trace pass's consumer (cbh_trace_batch, include/cerbos_hip.h): the evaluation_errors (field 7) and outputs (field 6)
 * of the CheckOutputs, from the log of a traced batch.  `b` = the batch cbi_flatten_pb made of the n traced inputs
 * (`bytes`, `offsets`), `res` = what cbh_trace_batch returned for it, `records` / `count` = its log.  Output i holds ONLY
 * those two fields, serialized: appended to the bytes cbi_assemble_pb produced for the same input they form one CheckOutput
 * (protobuf concatenation is a merge).  Errors come sorted and deduplicated (cel_errors.go:98-118), outputs in the order
 * check.go's loops reach them.  cbi_outputs_flags: CBI_TRACE_* where the device could not name everything - the decision
 * stands, errors / outputs of that input are the caller's engine's to supply. */
/* Which inputs want the trace pass: the ones whose output flags carry CBI_OUT_CEL_ERROR or CBI_OUT_WANTS_TRACE (the decision
 * kernels mark them, cerbos_hip.h CBH_ST_*).  cbi_table_trace_scope says what to expect of a table: 0 = the image has no trace
 * sections, 1 = only inputs with an absorbed error are ever marked, 2 = the table has variables (evaluated whether or not a
 * condition reads them, check.go:651-677) or rules with output expressions: the walk marks the inputs that visit such a rule
 * or whose variables fail, the older kernels (strict mode, requests wider than eight actions / four roles) every input. */
uint32_t cbi_table_trace_scope(const cbi_table* t);
#define CBI_TRACE_ERRORS_INCOMPLETE 4u
#define CBI_TRACE_OUTPUTS_INCOMPLETE 8u
int cbi_trace_pb(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint32_t* records, uint32_t count,
                 const uint8_t* bytes, const uint64_t* offsets, uint32_t n, cbi_outputs** out);
/* The same for the batch cbi_flatten_request_pb made of one CheckResourcesRequest (input i = its i-th resource entry), and the
 * response assembly that folds the traced outputs in: ResultEntry.outputs = CheckOutput.Outputs (cerbos_svc.go:325-327).
 * `traced` = what cbi_trace_request_pb returned, or NULL (= cbi_assemble_response_pb). */
int cbi_trace_request_pb(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint32_t* records, uint32_t count,
                         const uint8_t* request, uint64_t request_len, const uint8_t* aux_data, uint64_t aux_len, cbi_outputs** out);
int cbi_assemble_response_traced_pb(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint8_t* request,
                                    uint64_t request_len, const char* default_version, const cbi_outputs* traced, cbi_outputs** out);
void cbi_outputs_free(cbi_outputs* o);
/* Output i = bytes[offsets[i] .. offsets[i+1]); n + 1 offsets. */
const uint8_t* cbi_outputs_bytes(const cbi_outputs* o);
const uint64_t* cbi_outputs_offsets(const cbi_outputs* o);
/* Per input: CBI_OUT_* bits. */
const uint8_t* cbi_outputs_flags(const cbi_outputs* o);

#ifdef __cplusplus
}
#endif
#endif /* CERBOS_INGEST_H */
