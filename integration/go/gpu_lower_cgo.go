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
