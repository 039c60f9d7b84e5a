//go:build hipengine

// cgo binding of the lowering (include/cerbos_lower.h, INTEGRATION.md §1a): the runtimev1.RuleTable the manager has just built
// (internal/ruletable/manager.go:86-124) -> the device image newGPUEngine loads.  libcerbos_lower.so embeds the interpreter that
// runs the lowering; it is called once per published table, never on the path of a check.
package engine

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/cerbos_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/cerbos_hip/lib -lcerbos_lower
#include <stdlib.h>
#include "cerbos_lower.h"
*/
import "C"

import (
	"encoding/json"
	"errors"
	"fmt"
	"unsafe"

	"google.golang.org/protobuf/proto"

	runtimev1 "github.com/cerbos/cerbos/api/genpb/cerbos/runtime/v1"
)

// errCannotLower: the table holds constructs the device path refuses; the engine keeps it on the CPU path.
var errCannotLower = errors.New("gpu engine: rule table cannot be lowered")

// lowerRuleTable marshals the table and returns the device image.  globals == nil with perCallGlobals: `G.x` is read from the
// globals each call brings (evaluator.EvalParams.Globals); otherwise the configured globals become constants of the image.
func lowerRuleTable(rt *runtimev1.RuleTable, globals map[string]any, perCallGlobals bool) ([]byte, error) {
	pb, err := proto.MarshalOptions{Deterministic: true}.Marshal(rt)
	if err != nil {
		return nil, err
	}
	var cGlobals *C.char
	if globals != nil && !perCallGlobals {
		js, err := json.Marshal(globals)
		if err != nil {
			return nil, err
		}
		cGlobals = C.CString(string(js))
		defer C.free(unsafe.Pointer(cGlobals))
	}
	var flags C.uint32_t
	if perCallGlobals {
		flags |= C.CBL_PER_CALL_GLOBALS
	}
	var (
		image    *C.uint8_t
		imageLen C.size_t
		cErr     *C.char
		in       *C.uint8_t
	)
	if len(pb) > 0 {
		in = (*C.uint8_t)(unsafe.Pointer(&pb[0]))
	}
	st := C.cbl_lower_ruletable_pb(in, C.size_t(len(pb)), cGlobals, flags, &image, &imageLen, &cErr)
	if cErr != nil {
		defer C.cbl_free(unsafe.Pointer(cErr))
	}
	switch st {
	case C.CBL_OK:
		defer C.cbl_free(unsafe.Pointer(image))
		return C.GoBytes(unsafe.Pointer(image), C.int(imageLen)), nil
	case C.CBL_CANNOT_LOWER:
		return nil, fmt.Errorf("%w: %s", errCannotLower, C.GoString(cErr))
	default:
		return nil, fmt.Errorf("gpu engine: lowering failed (%d): %s", int(st), C.GoString(cErr))
	}
}

// ---- PlanResources (Engine.PlanResources, internal/engine/engine.go:141-170; ruletable/plan.go): the query planner is host code in the
// reference and stays host code here - it runs in the interpreter the lowering runs in (cerbos_amd/plan), one planner per published table.

type gpuPlanner struct{ h C.uint64_t }

// newPlanner: beside lowerRuleTable, when the manager publishes a table.
func newPlanner(rt *runtimev1.RuleTable) (*gpuPlanner, error) {
	pb, err := proto.MarshalOptions{Deterministic: true}.Marshal(rt)
	if err != nil {
		return nil, err
	}
	var h C.uint64_t
	var cErr *C.char
	var in *C.uint8_t
	if len(pb) > 0 {
		in = (*C.uint8_t)(unsafe.Pointer(&pb[0]))
	}
	if st := C.cbl_planner_open(in, C.size_t(len(pb)), &h, &cErr); st != C.CBL_OK {
		defer C.cbl_free(unsafe.Pointer(cErr))
		return nil, fmt.Errorf("gpu engine: planner (%d): %s", int(st), C.GoString(cErr))
	}
	return &gpuPlanner{h: h}, nil
}

func (p *gpuPlanner) close() { C.cbl_planner_close(p.h) }

// planParams is what Plan reads of evaluator.EvalParams.
type planParams struct {
	Globals              map[string]any `json:"globals,omitempty"`
	DefaultPolicyVersion string         `json:"defaultPolicyVersion,omitempty"`
	DefaultScope         string         `json:"defaultScope,omitempty"`
	LenientScopeSearch   bool           `json:"lenientScopeSearch,omitempty"`
	StrictEvaluation     bool           `json:"strictEvaluation,omitempty"`
	NowNs                int64          `json:"nowNs"`
}

// plan: proto.Marshal(input) in, the serialized enginev1.PlanResourcesOutput out (the caller unmarshals it; the AuditTrail's
// effective policies are not part of that message).
func (p *gpuPlanner) plan(inputPB []byte, params planParams) ([]byte, error) {
	js, err := json.Marshal(params)
	if err != nil {
		return nil, err
	}
	cParams := C.CString(string(js))
	defer C.free(unsafe.Pointer(cParams))
	var (
		out    *C.uint8_t
		outLen C.size_t
		cErr   *C.char
		in     *C.uint8_t
	)
	if len(inputPB) > 0 {
		in = (*C.uint8_t)(unsafe.Pointer(&inputPB[0]))
	}
	st := C.cbl_planner_plan_pb(p.h, in, C.size_t(len(inputPB)), cParams, &out, &outLen, &cErr)
	if cErr != nil {
		defer C.cbl_free(unsafe.Pointer(cErr))
	}
	if st != C.CBL_OK {
		return nil, fmt.Errorf("gpu engine: plan (%d): %s", int(st), C.GoString(cErr))
	}
	defer C.cbl_free(unsafe.Pointer(out))
	return C.GoBytes(unsafe.Pointer(out), C.int(outLen)), nil
}
