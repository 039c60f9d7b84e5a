/*
 * cerbos_hip.h - C ABI of libcerbos_hip.so, the MI355X-native batched decision engine for
 * the Cerbos CheckResources hot path.
 *
 * Drop-in seam: the reference evaluates one CheckInput at a time behind
 *     evaluator.Evaluator.Check(ctx, []*enginev1.CheckInput, ...CheckOpt)
 *         ([]*enginev1.CheckOutput, error)              internal/evaluator/evaluator.go:16-19
 * implemented by engine.(*Engine).Check (internal/engine/engine.go:216-240) which fans the
 * inputs out to ruletable.(*RuleTable).check (internal/ruletable/check.go:97-460).  This
 * library replaces the checkSerial/checkParallel fan-out (engine.go:225-229, 289-338): the
 * caller flattens a batch of CheckInputs into the SoA `cbh_batch`, the library evaluates
 * every (principal, resource, action) tuple on the GPU and returns per-tuple effect /
 * policy / scope ids the caller turns back into CheckOutput.Actions (check.go:64-94).
 * The cgo binding a Cerbos maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: plain C types only; every pointer in cbh_batch / cbh_result is HOST memory
 * owned by the caller (never retained after the call returns - cgo-safe).  Return value
 * 0 = OK, < 0 = hard error (text via cbh_last_error(), thread-local).  No function
 * aborts or throws across the boundary.  All entry points are thread-safe: one-shot calls
 * (cbh_check_batch) from different threads run on separate launch contexts and overlap on the
 * device (up to 8 per table and device, further callers wait); the resident calls on one batch
 * queue in call order (batches are dealt to a few streams per device, see cbh_table_set_resident_streams).
 *
 * Devices: cbh_init names the GPUs of the node the engine may use (the reference's fan-out over
 * NumCPU+4 goroutines, engine.go:309-338, becomes a fan-out over devices).  cbh_table_load puts
 * the image on the first device and broadcasts it to the others (RCCL over xGMI, peer copies if
 * RCCL is unavailable); cbh_check_batch shards a large batch into contiguous request ranges, one
 * per device, and writes every result at the caller's offsets (engine.go:332).
 */
#ifndef CERBOS_HIP_H
#define CERBOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBH_ABI_VERSION 3u
#define CBH_NONE 0xFFFFFFFFu

/* Effect values = effectv1.Effect (api/public/cerbos/effect/v1/effect.proto). */
#define CBH_EFFECT_ALLOW 1u
#define CBH_EFFECT_DENY 2u

/* cbh_params.flags - evaluator.EvalParams (internal/evaluator/evaluator.go:98-106). */
#define CBH_F_LENIENT_SCOPE_SEARCH 1u /* EvalParams.LenientScopeSearch */
#define CBH_F_STRICT_EVALUATION 2u    /* EvalParams.StrictEvaluation   */
#define CBH_F_WANT_DERIVED_ROLES 4u   /* fill cbh_result.edr_mask (CheckOutput.effective_derived_roles) */
#define CBH_F_WANT_EFFECTIVE_POLICIES 8u /* cbh_check_batch_trail: which policies' bindings were iterated (AuditTrail.EffectivePolicies); */
                                         /* the trail forms of the decision kernels */
#define CBH_F_DEBUG_CYCLES 0x100u     /* profiling aid: policy words carry per-wave cycle counts, not policies */

/* Per-request u32 fields, field-major: req_u32[field * n_requests + r]. */
enum cbh_req_field {
  CBH_RQ_PRINCIPAL_ID = 0, /* string id of principal.id                                   */
  CBH_RQ_P_SCOPE = 1,      /* scope index of the nearest table-known ancestor of the       */
                           /* effective principal scope; bit31 set = it IS that scope      */
  CBH_RQ_P_VERSION = 2,    /* string id of the effective principal policy version         */
  CBH_RQ_KIND = 3,         /* string id of namer.SanitizedResource(resource.kind)         */
  CBH_RQ_R_SCOPE = 4,      /* as P_SCOPE for the resource scope                           */
  CBH_RQ_R_VERSION = 5,    /* string id of the effective resource policy version          */
  CBH_RQ_ROLE_OFF = 6,     /* first entry of this principal's roles in `roles`            */
  CBH_RQ_ROLE_CNT = 7,
  CBH_RQ_ACT_OFF = 8,      /* this request's actions are tuple_action[ACT_OFF .. ACT_OFF+ACT_CNT)    */
  CBH_RQ_ACT_CNT = 9,      /* <= CBH_MAX_ACTIONS_PER_REQUEST; split larger CheckInputs into several */
  CBH_RQ_NCORE = 10,       /* fields 0..9 are read for every request; the rest are raw strings only CEL  */
                           /* programs read (uploaded only for a table that has such a program)       */
  CBH_RQ_S_RESOURCE_ID = 10,
  CBH_RQ_S_KIND = 11,
  CBH_RQ_S_P_SCOPE = 12,
  CBH_RQ_S_R_SCOPE = 13,
  CBH_RQ_S_P_VERSION = 14,
  CBH_RQ_S_R_VERSION = 15,
  CBH_RQ_NFIELDS = 16
};
#define CBH_MAX_ACTIONS_PER_REQUEST 64u /* the default request limit of the reference is 50 (server/conf.go:34-35) */
#define CBH_SCOPE_EXACT 0x80000000u

/* Attribute value tags (col_tag / heap_tag). */
enum cbh_tag {
  CBH_T_NULL = 0,
  CBH_T_BOOL = 1,   /* payload 0/1 */
  CBH_T_INT = 2,    /* int64 */
  CBH_T_UINT = 3,   /* uint64 */
  CBH_T_DOUBLE = 4, /* IEEE-754 binary64 bits; every JSON number arrives as this */
  CBH_T_STRING = 5, /* string id */
  CBH_T_LIST = 6,   /* payload = sel:2 | off:30 | len:32 ; elements at heap[off .. off+len) */
  CBH_T_MAP = 7,    /* payload as LIST; len pairs (key, value) at heap[off .. off+2*len)    */
  CBH_T_TIMESTAMP = 8, /* int64 ns since the Unix epoch */
  CBH_T_DURATION = 9,  /* int64 ns */
  CBH_T_ROPE = 10,     /* device only: a string a program put together (concatenation, lowerAscii / upperAscii as a value) - never  */
                       /* built, kept as the list of its parts: sel:2 (LOCAL) | off:30 | parts:32                                  */
  CBH_T_EDRSET = 11,   /* device only, trace programs: runtime.effectiveDerivedRoles AS A VALUE - payload = the derived-role mask; the host spells the sorted names */
  CBH_T_ABSENT = 0xF0, /* last key of a column path missing (has() -> false, read -> error) */
  CBH_T_ERR = 0xFF     /* reading the path is a CEL error (missing / non-map intermediate)  */
};
#define CBH_HEAP_TABLE 0u /* sel values */
#define CBH_HEAP_BATCH 1u
#define CBH_HEAP_ROLES 2u /* off/len index `roles`; elements are strings */
#define CBH_HEAP_LOCAL 3u /* device only: a list a program built (filter / map / intersect / except / +); never in a batch */

/* String usage flags (batch-local strings): which glob dimensions must be resolved. */
#define CBH_SF_ACTION 1u
#define CBH_SF_ROLE 2u
#define CBH_SF_KIND 4u

#define CBH_MAX_DEVICES 16
typedef struct cbh_config {
  uint32_t abi_version; /* CBH_ABI_VERSION */
  uint32_t n_devices;   /* entries of `devices` in use; 0 = every visible device */
  int32_t devices[CBH_MAX_DEVICES]; /* HIP device ordinals (a repeated ordinal = two replicas on one GPU: tests) */
} cbh_config;

/*
 * One batch of CheckInputs, flattened.  String ids: ids < cbh_table_num_strings() are the
 * table's own strings (the flattener interns against them so equality is id equality);
 * ids >= that are batch-local, index (id - num_table_strings) into str_off/str_flags.
 */
typedef struct cbh_batch {
  uint32_t n_requests;
  uint32_t n_tuples;
  uint32_t n_roles;   /* entries in `roles` */
  uint32_t n_columns; /* must equal cbh_table_num_columns() */
  uint32_t n_strings; /* batch-local strings */
  uint32_t heap_len;  /* entries in heap_tag / heap_val */
  uint64_t str_bytes_len;
  const uint32_t* req_u32;      /* [CBH_RQ_NFIELDS][n_requests] */
  const uint32_t* roles;        /* [n_roles] string ids */
  const uint32_t* tuple_req;    /* [n_tuples] request index (informational; tuples MUST be grouped by  */
                                /* request, each request owning the contiguous slice ACT_OFF/ACT_CNT) */
  const uint32_t* tuple_action; /* [n_tuples] string id of the action */
  const uint8_t* col_tag;       /* [n_columns][n_requests] */
  const uint64_t* col_val;      /* [n_columns][n_requests] */
  const uint8_t* heap_tag;      /* [heap_len] */
  const uint64_t* heap_val;     /* [heap_len] */
  const uint32_t* str_off;      /* [n_strings + 1] byte offsets into str_bytes */
  const uint8_t* str_bytes;     /* [str_bytes_len] UTF-8 */
  const uint8_t* str_flags;     /* [n_strings] CBH_SF_* */
} cbh_batch;

typedef struct cbh_params {
  int64_t now_ns; /* frozen clock of this Check call (evaluator_trace_common.go:24-26) */
  uint32_t flags; /* CBH_F_* */
  uint32_t reserved;
} cbh_params;

/* policy word = kind << 28 | id  (see cbh_policy_kind). */
enum cbh_policy_kind {
  CBH_P_EMPTY = 0,        /* ""  (principal without roles, check.go:191)                  */
  CBH_P_NO_MATCH = 1,     /* "NO_MATCH"                                                    */
  CBH_P_RESOURCE = 2,     /* resource.<kind>.v<version>[/scope]; id = scope index          */
  CBH_P_PRINCIPAL = 3,    /* principal.<id>.v<version>[/scope];  id = scope index          */
  CBH_P_TABLE = 4,        /* id = index into cbh_table policy-key list (role policies ...) */
  CBH_P_NO_MATCH_SCOPE_PERMISSIONS = 5
};

/* Per-tuple status. */
#define CBH_ST_OK 0u
#define CBH_ST_CEL_ERROR 1u   /* a CEL runtime error was absorbed (evaluation_errors non-empty) */
#define CBH_ST_UNSUPPORTED 2u /* hit an operation outside the device subset: result invalid;  */
                              /* the caller must evaluate this input with its own engine     */
#define CBH_ST_WANTS_TRACE 3u /* the decision stands and no CEL error was seen on the decision path, but what else the    */
                              /* reference reports for this input - outputs of a visited rule, the error of a variable    */
                              /* nothing reads - only cbh_trace_batch can tell.  Rule for callers: trace the inputs with a */
                              /* tuple marked CBH_ST_CEL_ERROR or CBH_ST_WANTS_TRACE.  (Kernels that cannot tell mark every */
                              /* tuple of a table with variables or outputs.)                                              */

typedef struct cbh_result {
  uint8_t* effect;    /* [n_tuples] CBH_EFFECT_*            (required) */
  uint32_t* policy;   /* [n_tuples] policy word             (optional, may be NULL) */
  uint32_t* scope;    /* [n_tuples] scope index or CBH_NONE (optional) */
  uint8_t* status;    /* [n_tuples] CBH_ST_*                (optional) */
  uint64_t* edr_mask; /* [n_requests] bit i = derived role i of the table's list (optional) */
} cbh_result;

typedef struct cbh_table cbh_table;           /* a lowered policy table resident on one GPU */
typedef struct cbh_device_batch cbh_device_batch; /* a batch resident in HBM */

int cbh_init(const cbh_config* cfg);
void cbh_shutdown(void);
const char* cbh_last_error(void);
uint32_t cbh_abi_version(void);

uint32_t cbh_num_devices(void);          /* devices the engine was initialised with */
int32_t cbh_device_ordinal(uint32_t i);  /* HIP ordinal of the i-th of them, -1 if out of range */

/* Page-locked host memory for batch / result arrays: cbh_check_batch moves arrays that live in such
 * memory by DMA straight from / to the caller's pages, chunked over several streams so that upload,
 * kernels and download overlap; ordinary (pageable) memory works too but goes through the driver's
 * staging copies.  C memory only, as cgo requires. */
void* cbh_alloc_pinned(size_t bytes);
void cbh_free_pinned(void* p);

/* Slabs.  A batch whose arrays lie in ONE page-locked block in the library's canonical order crosses PCIe in a
 * single copy (and a result slab comes back in one): cbh_batch_slab_bytes sizes the block from the counts in
 * `counts` (n_requests ... str_bytes_len), cbh_batch_bind_slab points the array members of `b` into it
 * (tuple_req = NULL: the device never reads it).  The producer then fills the arrays in place - this is how
 * libcerbos_ingest.so and the Go shim hand batches over.  Any other placement of the arrays is accepted too. */
size_t cbh_batch_slab_bytes(const cbh_batch* counts);
void cbh_batch_bind_slab(cbh_batch* b, void* slab);
size_t cbh_result_slab_bytes(uint32_t n_tuples, uint32_t n_requests);
void cbh_result_bind_slab(cbh_result* r, void* slab, uint32_t n_tuples, uint32_t n_requests);

/* Table lifetime.  The blob is the output of the lowering step (cerbos_amd.lower); it is copied to
 * the first device and broadcast to the others, the caller's buffer is not retained.
 * Tables are reference counted: cbh_table_load returns the owner's reference, every call that takes
 * a table holds one for its duration, cbh_table_retain adds one (a manager handing the current
 * table to a request, ruletable/manager.go:50-55), cbh_table_release drops one.  The table is freed
 * when the last reference goes - i.e. a released table drains its in-flight batches first
 * (manager.go:86-124: the swap does not wait for readers of the old table). */
int cbh_table_load(const void* blob, size_t len, cbh_table** out);
void cbh_table_retain(cbh_table* t);
void cbh_table_release(cbh_table* t);
/* How the image reached the other devices: "none" (one device), "rccl", "peer-copy". */
const char* cbh_table_broadcast_kind(const cbh_table* t);
uint32_t cbh_table_num_strings(const cbh_table* t);
uint32_t cbh_table_num_columns(const cbh_table* t);
uint64_t cbh_table_device_bytes(const cbh_table* t);
/* Device address of the table image (for a one-time RCCL broadcast to peer GPUs). */
void* cbh_table_device_ptr(const cbh_table* t);
/* Adopt an image that already sits in device memory (received by broadcast). */
int cbh_table_adopt_device_image(void* device_image, size_t len, cbh_table** out);

/* One-shot: upload `in`, evaluate, download into `out` (all host pointers).
 *  - a small batch (under 1 MB of inputs) is packed into a pinned staging block and crosses PCIe in one
 *    copy each way;
 *  - a large batch is split into contiguous request ranges, one per device (when the engine has several
 *    and the batch at least 16k requests), and each range is pipelined in chunks over three streams -
 *    upload of chunk c+1, kernels of chunk c and download of chunk c-1 overlap - provided the arrays are
 *    page-locked (cbh_alloc_pinned); pageable arrays are copied whole, array by array.
 * Requests must own contiguous, ascending tuple slices (ACT_OFF non-decreasing in request order) for
 * the split; any other order is accepted and evaluated in one piece on one device. */
int cbh_check_batch(cbh_table* t, const cbh_batch* in, const cbh_params* p, cbh_result* out);

/* Resident path (what bench.py times): inputs already in HBM when the clock starts.
 * cbh_batch_upload places the batch on the table's first device, cbh_batch_upload_on on the i-th. */
int cbh_batch_upload(cbh_table* t, const cbh_batch* in, cbh_device_batch** out);
int cbh_batch_upload_on(cbh_table* t, uint32_t device_index, const cbh_batch* in, cbh_device_batch** out);
void cbh_batch_release(cbh_device_batch* b);
/* Launches the kernels on the library's stream and returns without synchronising. */
int cbh_check_resident(cbh_table* t, cbh_device_batch* b, const cbh_params* p);
/* cbh_check_resident for bs[0 .. n) in order, in one call. */
int cbh_check_resident_many(cbh_table* t, cbh_device_batch* const* bs, uint32_t n, const cbh_params* p);
/* Resident batches are dealt round-robin to a few streams of their device at upload (a batch keeps its stream, so everything
 * that touches it stays ordered) and launches of batches on different streams overlap on the device: the dispatch ramp of one
 * fills the CUs the drain of another leaves idle.  n = 1 .. 8 for the batches uploaded from now on (default 4; 1 = strictly one
 * launch after the other - the setting for timing a kernel by itself). */
int cbh_table_set_resident_streams(cbh_table* t, uint32_t n);
uint32_t cbh_table_resident_streams(const cbh_table* t);
int cbh_synchronize(cbh_table* t); /* every device of the table */
/* The kernels cbh_check_resident launches for this batch and these parameters, e.g. "cbh_walk2_pre_kernel+cbh_walk2_kernel"
 * (thread-local string; a measurement aid). */
const char* cbh_plan_describe(cbh_table* t, cbh_device_batch* b, const cbh_params* p);
/* Copies the results of the last cbh_check_resident on `b` to host memory. */
int cbh_result_download(cbh_table* t, cbh_device_batch* b, cbh_result* out);
/* Average duration in ms of the decision kernel over the last `n` cbh_check_resident
 * launches, measured with HIP events on the library's own stream. */
int cbh_kernel_time_ms(cbh_table* t, float* check_kernel_ms, float* resolve_kernel_ms);

/* ---- Device-side ingest: serialized CheckInputs in, a resident batch out (the GPU flattens) ----------------------
 * The reference decodes each CheckInput and builds its request view on the CPU (internal/ruletable/check.go:536-554); so did
 * libcerbos_ingest.so (cbi_flatten_pb).  cbh_wire_flatten uploads the raw messages - `bytes`, message i =
 * bytes[offsets[i] .. offsets[i + 1]) - and three kernels (cerbos_amd/csrc/cbh_wire.h) build the batch in HBM: request words,
 * role / action ids, attribute columns, nested values, strings interned against the table and a batch-local dictionary.
 * The batch is then an ordinary resident batch: cbh_check_resident, cbh_result_download (results in INPUT order: tuple k of
 * input i at act_off[i] + k), cbh_batch_release.  Page-locked `bytes` (cbh_alloc_pinned) cross PCIe by DMA.
 * Returns 0; 1 = nothing was built because some messages (info->n_host of them; all, for a table with a column the device
 * flattener does not follow) are the host flattener's - more than 64 actions, a pre-0.30 resource kind no policy names,
 * containers nested deeper than 8: take the batch through cbi_flatten_pb + cbh_check_batch instead, same results;
 * < 0 = error (a malformed message: info->first_bad).
 * globals_pb / globals_len: the CALL's globals (evaluator.EvalParams.Globals, internal/evaluator/evaluator.go:52-57, 98-106) as
 * a serialized google.protobuf.Struct, or NULL / 0: a table lowered with per-call globals reads `G.x` from them (attribute columns
 * of root 4, like any request attribute); any other table carries its globals as constants of the image and ignores them. */
typedef struct cbh_wire_info {
  uint32_t n_requests; /* = n */
  uint32_t n_tuples;
  uint32_t n_host;     /* messages left to the host flattener (return value 1) */
  uint32_t first_bad;  /* index of the first malformed message, CBH_NONE */
  uint32_t dict_slots; /* slots of the batch-local string dictionary */
  uint32_t heap_len;   /* entries of the nested-value heap */
  uint32_t fill_runs;  /* 1, or more when the dictionary / heap had to grow */
  uint32_t n_routes;   /* > 0: the requests were grouped by route on the device (so many routes); 0: the batch is in input order */
} cbh_wire_info;
int cbh_wire_flatten(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                     const char* default_version, const char* default_scope, const uint8_t* globals_pb, size_t globals_len,
                     cbh_device_batch** out, cbh_wire_info* info);
/* Where the strings a CheckOutput repeats sit in each message (cerbos_ingest.h cbi_assemble_wire_pb reads them instead of
 * walking the messages again): in_span [n][6] (offset, length) pairs relative to the message start - request id, principal id,
 * principal version, resource kind, resource version, resource id; act_span [n_tuples] (offset, length) of every action;
 * act_off [n + 1] first tuple of every input. */
int cbh_wire_spans_download(cbh_table* t, cbh_device_batch* b, uint32_t* in_span, uint32_t* act_span, uint32_t* act_off);
/* ... or let the device write the answers too: after cbh_check_resident on a batch of cbh_wire_flatten, the serialized
 * enginev1.CheckOutput of every input - request_id, resource_id, actions -> {effect, policy, scope}, effective_derived_roles
 * (check.go:64-94, 513-530; byte for byte what cbi_assemble_wire_pb builds from the downloaded results) - back to back in
 * `bytes`, output i = bytes[offsets[i] .. offsets[i + 1]) (offsets: n + 1 entries), flags[i] = CBI_OUT_* of cerbos_ingest.h
 * (may be NULL).  Returns 0; 2 = `cap` is too small, *need holds the size (nothing was copied; call again); < 0 error.
 * evaluation_errors / outputs of the inputs whose flags ask for them come from the trace pass, as on the host road. */
int cbh_wire_outputs(cbh_table* t, cbh_device_batch* b, uint8_t* bytes, size_t cap, uint64_t* offsets, uint8_t* flags, size_t* need);
/* The whole device road in ONE call - what a Go caller's CheckBatch makes per batch (integration/go/gpu_cgo.go): serialized
 * CheckInputs in, serialized CheckOutputs out (replaces the loop of engine.go:289-338 over checkInputToRequest / check /
 * the CheckOutput marshalling of its callers).  The call is cut into up to four slices of contiguous messages that go down the
 * road side by side (a thread and a stream each), so that one slice's copies run under another's kernels: from ONE caller
 * thread this gives what several threads making the three calls in a row give.  out_offsets: n + 1 entries, output i =
 * out_bytes[out_offsets[i] .. out_offsets[i + 1]); out_flags (n entries, may be NULL) = CBI_OUT_* of cerbos_ingest.h.
 * Page-locked `bytes` / `out_bytes` (cbh_alloc_pinned) make every copy a DMA.  Returns 0; 1 = some message is the host
 * flattener's (info->n_host; nothing was written: take the host road); 2 = out_cap is too small, *need holds the size; < 0 error
 * (a malformed message: info->first_bad). */
int cbh_wire_check_pb(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                      const char* default_version, const char* default_scope, const uint8_t* globals_pb, size_t globals_len,
                      const cbh_params* p, uint8_t* out_bytes, size_t out_cap, uint64_t* out_offsets, uint8_t* out_flags,
                      size_t* need, cbh_wire_info* info);

/* ... and without blocking the caller.  cbh_wire_check_pb_submit starts the same call on a worker of the library and returns a
 * ticket; cbh_wire_check_pb_collect (the same table, or NULL; another table is refused and the ticket stays) waits for it, fills need / info and returns what cbh_wire_check_pb would have
 * returned (the error text of a failed call is the collecting thread's cbh_last_error).  The strings are copied at submit; every
 * buffer is the caller's and must stay valid and untouched until collect, which must be called exactly once per ticket.  ONE
 * caller thread that keeps two tickets in flight has the second call's uploads under the first's downloads: the fill and drain of a
 * call's slices are what a lone synchronous caller pays on top of the link's own time (bench.py `wire_inclusive_two_in_flight`).
 * For a cgo caller: a blocking C call pins an OS thread for its duration, this pair does not (engine.go:289-338's loop over
 * batches becomes submit, submit, collect, submit, collect ...). */
#define CBH_HAS_WIRE_CHECK_ASYNC 1
typedef struct cbh_wire_ticket cbh_wire_ticket;
int cbh_wire_check_pb_submit(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                             const char* default_version, const char* default_scope, const uint8_t* globals_pb, size_t globals_len,
                             const cbh_params* p, uint8_t* out_bytes, size_t out_cap, uint64_t* out_offsets, uint8_t* out_flags,
                             cbh_wire_ticket** ticket);
int cbh_wire_check_pb_collect(cbh_table* t, cbh_wire_ticket* ticket, size_t* need, cbh_wire_info* info);

/* ---- engine.Check's second return value (internal/engine/engine.go:217-240, 289-338): AuditTrail.EffectivePolicies ----
 * The reference records, for every binding its walk ITERATES, the source attributes of that binding's policy set
 * (check.go:302-304: the policy itself and, for a scoped resource / principal policy, its ancestors - compile.go:153-180) and
 * merges the inputs' trails per call.  cbh_check_batch_trail is cbh_check_batch plus that: effective_policies is
 * [n_groups][(cbh_table_num_policies + 31) / 32] words, bit p of group g set iff a binding of policy p was iterated for a request of
 * the group (group_of_request[i] < n_groups; NULL = every request in group 0: one engine.Check call).  cbh_table_policy_key names
 * policy p as namer.PolicyKeyFromFQN does ("resource.leave_request.vdefault/acme"); the keys a set bit stands for are that one
 * and, for a scoped resource / principal policy, those of its ancestor scopes that are policies of the table (drop the last scope
 * segment until none is left - cerbos_amd/engine.py effective_policy_keys is ten lines).  A flat table (resource policies only)
 * keeps its trail in the flat kernels (their trail instantiations sort out at the fold what a role BEHIND the allowing one touched);
 * a table of cbh_walk2_kernel's walks twice - once to learn the allowing roles, once over the walks the reference really makes,
 * marking; any other table is decided by the general walk, which iterates a request's roles one after the other as the reference
 * does.  Device 0. */
#define CBH_HAS_CHECK_BATCH_TRAIL 1
uint32_t cbh_table_num_policies(const cbh_table* t);
int cbh_table_policy_key(const cbh_table* t, uint32_t i, const char** key, uint32_t* len);
int cbh_check_batch_trail(cbh_table* t, const cbh_batch* in, const cbh_params* p, cbh_result* out, const uint32_t* group_of_request,
                          uint32_t n_groups, uint32_t* effective_policies);
/* ... for a RESIDENT batch (cbh_batch_upload*): cbh_batch_set_trail names the group of every request (host memory, the batch's device
 * order; NULL = one group) and clears the batch's masks; every cbh_check_resident with CBH_F_WANT_EFFECTIVE_POLICIES from then on ORs
 * into them; cbh_trail_download copies them out ([n_groups][words]).  cbh_check_batch_trail is upload + these + download. */
int cbh_batch_set_trail(cbh_table* t, cbh_device_batch* b, const uint32_t* group_of_request, uint32_t n_groups);
int cbh_trail_download(cbh_table* t, cbh_device_batch* b, uint32_t* effective_policies);

/* The device road for what the SERVER receives (internal/svc/cerbos_svc.go:255-344): `bytes` / `offsets` hold n_requests serialized
 * cerbos.request.v1.CheckResourcesRequest messages (request.proto:222-273).  Every resource entry becomes the CheckInput that
 * svc.CheckResources builds from it (cerbos_svc.go:274-288: the request's id and principal, the entry's resource and actions) - on
 * the device, by two launches in front of the flattener (cbh_wire_req.h); the host never sees those messages.  aux_bytes /
 * aux_offsets (n_requests + 1 entries; both NULL = none): per request the serialized cerbos.engine.v1.AuxData the server derived
 * from the request's JWT (auxdata.Extract, cerbos_svc.go:262 - verification is the caller's business), empty where there is none.
 * first_input [n_requests + 1]: the inputs (and, after the check, outputs) of request r are first_input[r] .. first_input[r + 1];
 * request_flags [n_requests] (may be NULL): bit 0 = include_meta.  info->n_requests = the number of inputs; info->first_bad
 * indexes requests.  The batch goes through cbh_check_resident / cbh_wire_outputs like one of cbh_wire_flatten (spans downloaded
 * with cbh_wire_spans_download refer to the device-made messages, which the host does not have: use cbh_wire_outputs).
 * Same last-field-wins grammar as cerbos_ingest.h cbi_flatten_request_pb, the host road for one request. */
int cbh_wire_flatten_requests(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n_requests,
                              const uint8_t* aux_bytes, const uint64_t* aux_offsets, const char* default_version, const char* default_scope,
                              const uint8_t* globals_pb, size_t globals_len, uint32_t* first_input, uint8_t* request_flags,
                              cbh_device_batch** out, cbh_wire_info* info);
/* ... and bytes in, bytes out in one call: the serialized CheckOutputs (what engine.Check returns, engine.go:217-240) of every
 * resource entry of every request, back to back; out_offsets (out_inputs_cap + 1 entries) / out_flags (out_inputs_cap entries, may
 * be NULL) as cbh_wire_outputs.  Returns as cbh_wire_check_pb; 2 = out_cap or out_inputs_cap is too small: *need holds the bytes,
 * info->n_requests the inputs (nothing was written; call again). */
int cbh_wire_check_requests_pb(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n_requests,
                               const uint8_t* aux_bytes, const uint64_t* aux_offsets, const char* default_version, const char* default_scope,
                               const uint8_t* globals_pb, size_t globals_len, const cbh_params* p, uint32_t* first_input, uint8_t* request_flags,
                               uint8_t* out_bytes, size_t out_cap, uint64_t* out_offsets, uint8_t* out_flags, size_t out_inputs_cap, size_t* need,
                               cbh_wire_info* info);
/* ... with the audit trail of every request beside its outputs: effective_policies [n_requests][words], words =
 * (cbh_table_num_policies + 31) / 32 - bit k of row r set when policy k (cbh_table_policy_key) is among those the engine went
 * through for any resource entry of request r: AuditTrail.EffectivePolicies (check.go:302-304) of the one decision-log entry
 * the server writes for a CheckResources call.  The trail kernels of cbh_check_batch_trail with one group per request. */
int cbh_wire_check_requests_trail_pb(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n_requests,
                                     const uint8_t* aux_bytes, const uint64_t* aux_offsets, const char* default_version, const char* default_scope,
                                     const uint8_t* globals_pb, size_t globals_len, const cbh_params* p, uint32_t* first_input, uint8_t* request_flags,
                                     uint8_t* out_bytes, size_t out_cap, uint64_t* out_offsets, uint8_t* out_flags, size_t out_inputs_cap, size_t* need,
                                     cbh_wire_info* info, uint32_t* effective_policies);

/* ---- Trace pass: evaluation_errors and outputs (evaluator/cel_errors.go:48-118, check.go:383-411, 776-807) ----
 * The decision kernels only mark the tuples whose evaluation absorbed a CEL error (CBH_ST_CEL_ERROR).  What the
 * reference reports beyond the effect - the (expression, message) pairs and the values of the rules' output
 * expressions - comes from a second launch over the inputs that need it (the ones with that mark, or every input of
 * a table with outputs): the same walk, with programs that keep the expression identities, writing fixed-size
 * records to a log.  cbh_trace_batch decides `in` again (its result must equal cbh_check_batch's) and fills the log.
 *
 * A record is eight 32-bit words:
 *   w0  request index (into `in`)
 *   w1  kind (bits 0-3) | pass << 4 (0 principal policies, 1 resource policies) | (bit 5, see w4) | part << 6 |
 *       role iteration << 12 | site << 20
 *       (site = position of the rule record in this request's walk of that pass; outputs are ordered by action, pass,
 *        role iteration, site as check.go's loops nest)
 *   w2  trace string id: the expression text (errors), or the rule's FQN (outputs)   [CBH_SEC_TRACE_STRINGS]
 *   w3  CBH_TR_ERROR / CBH_TR_OUTPUT_ERROR: error code | detail << 8;  CBH_TR_OUTPUT: value tag (cbh_value_tag) | rule << 8
 *   w4, w5  CBH_TR_ERROR: the whole error payload (code | detail << 8, 64 bits); CBH_TR_OUTPUT: the value (64 bits; strings are string ids, lists / maps refer to the batch heap);
 *           CBH_TR_OUTPUT_ERROR: w4 = rule.   (part: what an output expression builds - list / map literals, format() -
 *           is assembled by the consumer from a template the lowering keeps; the device evaluates the computed parts below
 *           those constructors and logs one record per part, numbered in evaluation order; an expression without such
 *           constructors is part 0.  rule = a small id per evaluation key, ruletable.go's EvaluationKey, bit 23 set for
 *           the conditionNotMet expression: w1 bit 5
 *           marks a visit whose derived-role condition failed - check.go:343-347 emits nothing on the FIRST such visit
 *           of a key and conditionNotMet on later ones, which the consumer replays)
 *   w6, w7  CBH_TR_OUTPUT*: mask of the request's actions the rule was visited for (bit k = k-th action)
 * count may exceed capacity: the log overflowed, call again with a larger one. */
#define CBH_TR_ERROR 1u         /* a condition / variable expression failed */
#define CBH_TR_OUTPUT 2u        /* an output expression produced a value */
#define CBH_TR_OUTPUT_ERROR 3u  /* an output expression failed: OutputEntry.error */
#define CBH_TR_OUTPUT_ELEMENT 5u /* element w6 of the list the preceding CBH_TR_OUTPUT record of this visit / part refers to with   */
                                /* CBH_HEAP_LOCAL (a list the program built: it lives in the lane's LDS, so its elements are logged) */
#define CBH_TR_INCOMPLETE 4u    /* an output expression of this request is outside the device subset: its outputs are not all */
                                /* here.  (A condition / variable outside the subset marks the tuples CBH_ST_UNSUPPORTED.)     */
/* error codes (w3 bits 0-7); detail = w3 >> 8 */
#define CBH_ERR_OTHER 0u            /* an error the device does not classify: the message is not available */
#define CBH_ERR_NO_SUCH_KEY 1u      /* "no such key: <string detail>"            (detail = string id) */
#define CBH_ERR_ATTR_MISSING 2u     /* an attribute path did not resolve; detail = column index: "no such key: <first missing */
                                    /* key of the path>" or, below a non-map value, "no such overload"                        */
#define CBH_ERR_NO_SUCH_OVERLOAD 3u /* "no such overload" */
#define CBH_ERR_UNDEFINED_FIELD 4u  /* "undefined field '<name>'"                (detail = trace string id of the variable) */
#define CBH_ERR_DIV_BY_ZERO 5u      /* "division by zero" */
#define CBH_ERR_MOD_BY_ZERO 6u      /* "modulus by zero" */
#define CBH_ERR_INT_OVERFLOW 7u     /* "integer overflow" */
#define CBH_ERR_UINT_OVERFLOW 8u    /* "unsigned integer overflow" */
#define CBH_ERR_EDR_FAILED 9u       /* strict mode: "failed to compute effective derived roles [a, b]"; the 56-bit detail (w4, w5 = */
                                    /* the whole payload, code in the low byte) = mask of the failed roles, bit = edr_mask bit       */
#define CBH_TRACE_RECORD_WORDS 8u

typedef struct cbh_trace {
  uint32_t* records;  /* [capacity][CBH_TRACE_RECORD_WORDS] */
  uint32_t capacity;  /* records that fit */
  uint32_t count;     /* out: records the pass produced */
} cbh_trace;

/* Decide `in` with the tracing kernel on the table's first device: `out` as cbh_check_batch, `trace` as above. */
int cbh_trace_batch(cbh_table* t, const cbh_batch* in, const cbh_params* p, cbh_result* out, cbh_trace* trace);

#ifdef __cplusplus
}
#endif
#endif /* CERBOS_HIP_H */
