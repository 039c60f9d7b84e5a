/* cerbos_lower.h - the lowering behind a C ABI: serialized runtimev1.RuleTable in, device table image out.
 *
 * Replaces, for a Go host, what `ruletable.NewRuleTable` does with a freshly compiled table before it serves checks
 * (internal/ruletable/ruletable.go:637-691; rebuilt on every storage event by ruletable.Manager, manager.go:86-124): the host
 * marshals its runtimev1.RuleTable (api/private/cerbos/runtime/v1/runtime.proto:41-105; private/ruletable/ruletable.go:27-44),
 * calls cbl_lower_ruletable_pb and hands the image to cbh_table_load (cerbos_hip.h) / cbi_table_open (cerbos_ingest.h).
 *
 * libcerbos_lower.so carries no lowering of its own: it embeds the CPython interpreter (libpython3.10) and runs the package's
 * lowering (cerbos_amd/lower: CEL parser and compiler, regex automata, glob tables, constant folder, time-zone tables) in the
 * calling process - the same code as `python -m cerbos_amd.lower`, without a child process.  It runs once per published table,
 * never on the path of a check.  The package is found next to the library (<dir>/../cerbos_amd) or under $CERBOS_AMD_ROOT.
 * Calls are serialised on the interpreter's lock; any thread may call.
 */
#ifndef CERBOS_LOWER_H
#define CERBOS_LOWER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBL_ABI_VERSION 1

/* flags of cbl_lower_ruletable_pb */
#define CBL_PER_CALL_GLOBALS 1u /* `G.x` is read from the globals every call brings (cbh_wire_flatten / cbi_flatten_pb_g): one image for any globals */
#define CBL_NO_TRACE 2u         /* leave the trace pass's sections out: decisions only, no evaluation_errors / outputs */

/* status */
#define CBL_OK 0
#define CBL_CANNOT_LOWER 2 /* the table holds constructs the device path refuses (the caller keeps it on its CPU engine); *error says which */
#define CBL_BAD_INPUT 3    /* not a runtimev1.RuleTable, bad globals JSON */
#define CBL_RUNTIME 4      /* the interpreter or the package could not be started; *error says why */

int cbl_abi_version(void);

/* ruletable_pb / len : proto.Marshal of the runtimev1.RuleTable
 * globals_json       : NULL, or a JSON object - the engine's configured globals (evaluator/conf.go:40), folded into the image
 *                      unless CBL_PER_CALL_GLOBALS
 * image / image_len  : on CBL_OK the image, allocated by the library: release it with cbl_free
 * error              : on any other status a NUL-terminated message, release with cbl_free (may be NULL: no message wanted)   */
int cbl_lower_ruletable_pb(const uint8_t* ruletable_pb, size_t len, const char* globals_json, uint32_t flags,
                           uint8_t** image, size_t* image_len, char** error);

/* The same, and the lowering's statistics with it: *stats_json (may be NULL) = one JSON object (sizes, the kernels the table is eligible
 * for, what is outside the device subset), NUL-terminated, release with cbl_free.  What a caller whose threads are not its own (a Go
 * program: a goroutine may resume on another OS thread) uses instead of cbl_last_stats_json. */
int cbl_lower_ruletable_pb_stats(const uint8_t* ruletable_pb, size_t len, const char* globals_json, uint32_t flags,
                                 uint8_t** image, size_t* image_len, char** stats_json, char** error);

/* the lowering's statistics of the LAST successful call on this thread's behalf, one JSON object (sizes, kernels the table is
 * eligible for, what is outside the device subset); NUL-terminated, release with cbl_free; NULL before the first call */
char* cbl_last_stats_json(void);

void cbl_free(void* p);

/* ---- PlanResources (internal/ruletable/plan.go:31-415, internal/ruletable/planner): the query planner behind the same library.
 * Host-side and symbolic in the reference too - one principal, actions and a resource KIND in, the condition under which a resource
 * of that kind is allowed out - so it runs where the lowering runs (cerbos_amd/plan in the embedded interpreter), not on the GPU.
 * A planner belongs to a published rule table: open it when the table is published (beside cbl_lower_ruletable_pb), plan with it
 * from any thread (calls are serialised on the interpreter's lock), close it when the table is replaced.
 *   input_pb    : proto.Marshal of the enginev1.PlanResourcesInput (api/public/cerbos/engine/v1/engine.proto:20-60)
 *   params_json : NULL, or the call's evaluator parameters as a JSON object - "globals" {..}, "defaultPolicyVersion",
 *                 "defaultScope", "lenientScopeSearch", "strictEvaluation", "nowNs" (evaluator.EvalParams)
 *   output_pb   : on CBL_OK the serialized enginev1.PlanResourcesOutput (engine.proto:116-128: filter, filter_debug, matched_scopes,
 *                 evaluation_errors), allocated by the library: release it with cbl_free */
int cbl_planner_open(const uint8_t* ruletable_pb, size_t len, uint64_t* planner, char** error);
int cbl_planner_plan_pb(uint64_t planner, const uint8_t* input_pb, size_t len, const char* params_json, uint8_t** output_pb, size_t* output_len,
                        char** error);
void cbl_planner_close(uint64_t planner);

#ifdef __cplusplus
}
#endif
#endif
