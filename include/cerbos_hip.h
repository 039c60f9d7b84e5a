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
 * device (up to 8 per table, further callers wait); the resident calls of one table share one
 * stream and queue in call order.
 */
#ifndef CERBOS_HIP_H
#define CERBOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBH_ABI_VERSION 2u
#define CBH_NONE 0xFFFFFFFFu

/* Effect values = effectv1.Effect (api/public/cerbos/effect/v1/effect.proto). */
#define CBH_EFFECT_ALLOW 1u
#define CBH_EFFECT_DENY 2u

/* cbh_params.flags - evaluator.EvalParams (internal/evaluator/evaluator.go:98-106). */
#define CBH_F_LENIENT_SCOPE_SEARCH 1u /* EvalParams.LenientScopeSearch */
#define CBH_F_STRICT_EVALUATION 2u    /* EvalParams.StrictEvaluation   */
#define CBH_F_WANT_DERIVED_ROLES 4u   /* fill cbh_result.edr_mask (CheckOutput.effective_derived_roles) */
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
  CBH_RQ_S_RESOURCE_ID = 8,  /* the rest are raw strings only CEL programs read           */
  CBH_RQ_S_KIND = 9,
  CBH_RQ_S_P_SCOPE = 10,
  CBH_RQ_S_R_SCOPE = 11,
  CBH_RQ_S_P_VERSION = 12,
  CBH_RQ_S_R_VERSION = 13,
  CBH_RQ_ACT_OFF = 14,     /* this request's actions are tuple_action[ACT_OFF .. ACT_OFF+ACT_CNT)    */
  CBH_RQ_ACT_CNT = 15,     /* <= CBH_MAX_ACTIONS_PER_REQUEST; split larger CheckInputs into several */
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
  CBH_T_ABSENT = 0xF0, /* last key of a column path missing (has() -> false, read -> error) */
  CBH_T_ERR = 0xFF     /* reading the path is a CEL error (missing / non-map intermediate)  */
};
#define CBH_HEAP_TABLE 0u /* sel values */
#define CBH_HEAP_BATCH 1u
#define CBH_HEAP_ROLES 2u /* off/len index `roles`; elements are strings */

/* String usage flags (batch-local strings): which glob dimensions must be resolved. */
#define CBH_SF_ACTION 1u
#define CBH_SF_ROLE 2u
#define CBH_SF_KIND 4u

typedef struct cbh_config {
  uint32_t abi_version; /* CBH_ABI_VERSION */
  int32_t device;       /* HIP device ordinal */
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

/* Table lifetime.  The blob is the output of the lowering step (cerbos_amd.lower); it is
 * copied to the device, the caller's buffer is not retained. */
int cbh_table_load(const void* blob, size_t len, cbh_table** out);
void cbh_table_release(cbh_table* t);
uint32_t cbh_table_num_strings(const cbh_table* t);
uint32_t cbh_table_num_columns(const cbh_table* t);
uint64_t cbh_table_device_bytes(const cbh_table* t);
/* Device address of the table image (for a one-time RCCL broadcast to peer GPUs). */
void* cbh_table_device_ptr(const cbh_table* t);
/* Adopt an image that already sits in device memory (received by broadcast). */
int cbh_table_adopt_device_image(void* device_image, size_t len, cbh_table** out);

/* One-shot: upload `in`, evaluate, download into `out` (all host pointers).  A batch under 4 MB is packed into a
 * pinned staging block and crosses PCIe in one copy each way. */
int cbh_check_batch(cbh_table* t, const cbh_batch* in, const cbh_params* p, cbh_result* out);

/* Resident path (what bench.py times): inputs already in HBM when the clock starts. */
int cbh_batch_upload(cbh_table* t, const cbh_batch* in, cbh_device_batch** out);
void cbh_batch_release(cbh_device_batch* b);
/* Launches the kernels on the library's stream and returns without synchronising. */
int cbh_check_resident(cbh_table* t, cbh_device_batch* b, const cbh_params* p);
int cbh_synchronize(cbh_table* t);
/* Copies the results of the last cbh_check_resident on `b` to host memory. */
int cbh_result_download(cbh_table* t, cbh_device_batch* b, cbh_result* out);
/* Average duration in ms of the decision kernel over the last `n` cbh_check_resident
 * launches, measured with HIP events on the library's own stream. */
int cbh_kernel_time_ms(cbh_table* t, float* check_kernel_ms, float* resolve_kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* CERBOS_HIP_H */
