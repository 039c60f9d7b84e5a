//go:build hipengine

// cgo binding of the MI355X decision engine behind engine.Check (INTEGRATION.md §1-2a).  Bytes in, bytes out:
// the CheckInputs are marshalled once, libcerbos_ingest.so flattens them, libcerbos_hip.so decides, and
// libcerbos_ingest.so assembles the serialized CheckOutputs.  Inputs the device cannot evaluate
// (CBI_OUT_UNSUPPORTED) are handed back to the caller's CPU path, which also serves the whole batch on any error.
package engine

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/cerbos_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/cerbos_hip/lib -lcerbos_ingest -lcerbos_hip
#include <stdlib.h>
#include "cerbos_ingest.h"
*/
import "C"

import (
	"context"
	"errors"
	"fmt"
	"runtime"
	"sort"
	"strings"
	"sync"
	"unsafe"

	"google.golang.org/protobuf/proto"

	enginev1 "github.com/cerbos/cerbos/api/genpb/cerbos/engine/v1"
	"github.com/cerbos/cerbos/internal/evaluator"
)

type gpuEngine struct {
	table  *C.cbh_table // device image of the lowered rule table
	ingest *C.cbi_table // host dictionaries of the same image (immutable: shared by all goroutines)
	outPool pinnedPool  // page-locked blocks the device road's answers land in
}

// devices: the HIP ordinals the engine may use. A large batch is cut into one contiguous request range per device
// (the fan-out of engine.go:309-338 with GPUs for workers); the image is broadcast once at load.
func newGPUEngine(blob []byte, devices []int) (*gpuEngine, error) {
	if len(blob) == 0 || len(devices) > C.CBH_MAX_DEVICES {
		return nil, errors.New("gpu engine: empty table image or too many devices")
	}
	cfg := C.cbh_config{abi_version: C.CBH_ABI_VERSION, n_devices: C.uint32_t(len(devices))}
	for i, d := range devices {
		cfg.devices[i] = C.int32_t(d)
	}
	if C.cbh_init(&cfg) != 0 {
		return nil, errors.New(C.GoString(C.cbh_last_error()))
	}
	g := &gpuEngine{}
	// both libraries copy the image: a Go pointer may be passed for the duration of the call
	if C.cbh_table_load(unsafe.Pointer(&blob[0]), C.size_t(len(blob)), &g.table) != 0 {
		return nil, errors.New(C.GoString(C.cbh_last_error()))
	}
	if C.cbi_table_open(unsafe.Pointer(&blob[0]), C.size_t(len(blob)), &g.ingest) != 0 {
		C.cbh_table_release(g.table)
		return nil, errors.New(C.GoString(C.cbi_last_error()))
	}
	return g, nil
}

// close drops the owner's reference: the image leaves HBM when the last in-flight batch has drained
// (ruletable/manager.go:86-124 swaps tables without waiting for readers; so does this).
func (g *gpuEngine) close() {
	C.cbi_table_close(g.ingest)
	C.cbh_table_release(g.table)
}

// checkBatchGPU returns one output per input; fallback[i] is true where the caller must run input i itself.
func (g *gpuEngine) checkBatchGPU(_ context.Context, inputs []*enginev1.CheckInput, p evaluator.EvalParams) (outs []*enginev1.CheckOutput, fallback []bool, err error) {
	n := len(inputs)
	if n == 0 { // nothing to pin: &buf[0] of an empty slice panics
		return nil, nil, nil
	}
	buf := make([]byte, 0, 256*n)
	offs := make([]C.uint64_t, 1, n+1)
	for _, in := range inputs {
		if buf, err = (proto.MarshalOptions{}).MarshalAppend(buf, in); err != nil {
			return nil, nil, err
		}
		offs = append(offs, C.uint64_t(len(buf)))
	}
	if len(buf) == 0 { // n all-empty messages serialise to nothing: keep &buf[0] valid
		buf = append(buf, 0)[:1]
	}
	var pin runtime.Pinner // the C side only reads these during the calls below
	pin.Pin(&buf[0])
	pin.Pin(&offs[0])
	defer pin.Unpin()
	bytesPtr := (*C.uint8_t)(unsafe.Pointer(&buf[0]))

	dv, ds := C.CString(p.DefaultPolicyVersion), C.CString(p.DefaultScope)
	defer C.free(unsafe.Pointer(dv))
	defer C.free(unsafe.Pointer(ds))

	flags := C.uint32_t(C.CBH_F_WANT_DERIVED_ROLES)
	if p.LenientScopeSearch {
		flags |= C.CBH_F_LENIENT_SCOPE_SEARCH
	}
	if p.StrictEvaluation {
		flags |= C.CBH_F_STRICT_EVALUATION
	}
	params := C.cbh_params{now_ns: C.int64_t(p.NowFunc().UnixNano()), flags: flags}

	// The device road first (INTEGRATION.md §2d): the raw bytes cross PCIe once, the GPU flattens them, decides and writes
	// the serialized CheckOutputs.  What the device flattener leaves (a CheckInput with more than 64 actions, ...) takes
	// the host road below - same bytes either way.
	obytes, ooffs, oflags, release, ok, err := g.deviceRoad(bytesPtr, offs, n, dv, ds, &params)
	if err != nil {
		return nil, nil, err
	}
	if !ok {
		if obytes, ooffs, oflags, release, err = g.hostRoad(bytesPtr, offs, n, dv, ds, &params); err != nil {
			return nil, nil, err
		}
	}
	defer release()

	// evaluation_errors / outputs (check.go:90-92): the inputs that can have any go through the tracing kernel once more
	// (INTEGRATION.md §2c); what comes back per input is just those two fields, serialized - merged into the output below.
	extra, err := g.traceGPU(buf, offs, oflags, n, dv, ds, &params)
	if err != nil {
		return nil, nil, err
	}

	outs = make([]*enginev1.CheckOutput, n)
	fallback = make([]bool, n)
	for i := range inputs {
		if oflags[i]&C.CBI_OUT_UNSUPPORTED != 0 {
			fallback[i] = true
			continue
		}
		out := &enginev1.CheckOutput{}
		if err := proto.Unmarshal(obytes[ooffs[i]:ooffs[i+1]], out); err != nil {
			return nil, nil, fmt.Errorf("output %d: %w", i, err)
		}
		if x, ok := extra[i]; ok {
			if x.incomplete { // the device could not name every error / output of this input: the CPU path does
				fallback[i] = true
				continue
			}
			if err := (proto.UnmarshalOptions{Merge: true}).Unmarshal(x.bytes, out); err != nil {
				return nil, nil, fmt.Errorf("trace of output %d: %w", i, err)
			}
		}
		outs[i] = out
	}
	return outs, fallback, nil
}

// hostRoad: cbi_flatten_pb -> cbh_check_batch -> cbi_assemble_pb (a host thread parses and assembles).  The returned slices
// alias C memory until release() is called.
func (g *gpuEngine) hostRoad(bytesPtr *C.uint8_t, offs []C.uint64_t, n int, dv, ds *C.char, params *C.cbh_params) (obytes []byte, ooffs []uint64, oflags []byte, release func(), err error) {
	var batch *C.cbi_batch
	if C.cbi_flatten_pb(g.ingest, bytesPtr, &offs[0], C.uint32_t(n), dv, ds, 1, &batch) != 0 {
		return nil, nil, nil, nil, errors.New(C.GoString(C.cbi_last_error()))
	}
	defer C.cbi_batch_free(batch)
	view := C.cbi_batch_view(batch)

	// result arrays in C memory (cgo: no Go pointers to Go pointers inside cbh_result)
	nt, nr := C.size_t(view.n_tuples), C.size_t(view.n_requests)
	res := C.cbh_result{
		effect:   (*C.uint8_t)(C.calloc(nt+1, 1)),
		policy:   (*C.uint32_t)(C.calloc(nt+1, 4)),
		scope:    (*C.uint32_t)(C.calloc(nt+1, 4)),
		status:   (*C.uint8_t)(C.calloc(nt+1, 1)),
		edr_mask: (*C.uint64_t)(C.calloc(nr+1, 8)),
	}
	defer func() {
		C.free(unsafe.Pointer(res.effect))
		C.free(unsafe.Pointer(res.policy))
		C.free(unsafe.Pointer(res.scope))
		C.free(unsafe.Pointer(res.status))
		C.free(unsafe.Pointer(res.edr_mask))
	}()

	if C.cbh_check_batch(g.table, view, params, &res) != 0 {
		return nil, nil, nil, nil, errors.New(C.GoString(C.cbh_last_error()))
	}

	var assembled *C.cbi_outputs
	if C.cbi_assemble_pb(g.ingest, batch, &res, bytesPtr, &offs[0], C.uint32_t(n), dv, &assembled) != 0 {
		return nil, nil, nil, nil, errors.New(C.GoString(C.cbi_last_error()))
	}
	ooffs = unsafe.Slice((*uint64)(unsafe.Pointer(C.cbi_outputs_offsets(assembled))), n+1)
	oflags = unsafe.Slice((*byte)(unsafe.Pointer(C.cbi_outputs_flags(assembled))), n)
	obytes = unsafe.Slice((*byte)(unsafe.Pointer(C.cbi_outputs_bytes(assembled))), int(ooffs[n]))

	return obytes, ooffs, oflags, func() { C.cbi_outputs_free(assembled) }, nil
}

// deviceRoad: cbh_wire_check_pb (= cbh_wire_flatten -> cbh_check_resident -> cbh_wire_outputs, sliced).  ok == false: these messages are the host
// flattener's (the call returned 1).  The answers land in a page-locked block from g.outPool (DMA, no staging copy).
func (g *gpuEngine) deviceRoad(bytesPtr *C.uint8_t, offs []C.uint64_t, n int, dv, ds *C.char, params *C.cbh_params) (obytes []byte, ooffs []uint64, oflags []byte, release func(), ok bool, err error) {
	// cbh_wire_check_pb: the whole road in one cgo crossing - the library cuts the call into slices that go down side by side
	// (one slice's copies under another's kernels), so ONE goroutine gets what several would making the three calls in a row
	blk := g.outPool.get(192*n+4096, n) // bytes | offsets (n + 1) | flags (n), one page-locked block
	if blk == nil {                     // no page-locked memory to be had: the host road answers this call
		return nil, nil, nil, nil, false, nil
	}
	var need C.size_t
	var info C.cbh_wire_info
	call := func() C.int {
		return C.cbh_wire_check_pb(g.table, 0, bytesPtr, &offs[0], C.uint32_t(n), dv, ds, nil, 0 /* per-call globals: a serialized Struct */, params,
			blk.bytes, C.size_t(blk.cap), blk.offs, blk.flags, &need, &info)
	}
	rc := call()
	if rc == 2 { // the guess was short: `need` is exact
		g.outPool.put(blk)
		if blk = g.outPool.get(int(need), n); blk == nil {
			return nil, nil, nil, nil, false, nil
		}
		rc = call()
	}
	switch {
	case rc == 1: // some message is the host flattener's (more than 64 actions, ...): the host road takes the call
		g.outPool.put(blk)
		return nil, nil, nil, nil, false, nil
	case rc != 0:
		g.outPool.put(blk)
		return nil, nil, nil, nil, false, errors.New(C.GoString(C.cbh_last_error()))
	}
	ooffs = unsafe.Slice((*uint64)(unsafe.Pointer(blk.offs)), n+1)
	oflags = unsafe.Slice((*byte)(unsafe.Pointer(blk.flags)), n)
	obytes = unsafe.Slice((*byte)(unsafe.Pointer(blk.bytes)), int(ooffs[n]))
	return obytes, ooffs, oflags, func() { g.outPool.put(blk) }, true, nil
}

// deviceRoadAsync: the same road without holding an OS thread for the call's duration (a blocking cgo call pins one):
// cbh_wire_check_pb_submit starts the call on a worker of the library, the returned function collects it.  A server loop that keeps
// two of these in flight per GPU (submit, submit, collect, submit, collect ...) has one call's uploads under the other's downloads.
// The input block and `blk` must stay untouched until collect; C memory only (cgo's pointer rules: no Go pointer outlives the call
// that passed it - bytesPtr / offs must therefore be C allocations here, as blk is).
func (g *gpuEngine) deviceRoadAsync(bytesPtr *C.uint8_t, offs *C.uint64_t, n int, dv, ds *C.char, params *C.cbh_params, blk *pinnedBlock) (collect func() (obytes []byte, ooffs []uint64, oflags []byte, rc int, err error), err error) {
	var ticket *C.cbh_wire_ticket
	if C.cbh_wire_check_pb_submit(g.table, 0, bytesPtr, offs, C.uint32_t(n), dv, ds, nil, 0, params, blk.bytes, C.size_t(blk.cap), blk.offs, blk.flags, &ticket) != 0 {
		return nil, errors.New(C.GoString(C.cbh_last_error()))
	}
	return func() ([]byte, []uint64, []byte, int, error) {
		var need C.size_t
		var info C.cbh_wire_info
		rc := int(C.cbh_wire_check_pb_collect(g.table, ticket, &need, &info))
		if rc < 0 {
			return nil, nil, nil, rc, errors.New(C.GoString(C.cbh_last_error()))
		}
		if rc != 0 { // 1: the host flattener's; 2: the block was short (need bytes)
			return nil, nil, nil, rc, nil
		}
		ooffs := unsafe.Slice((*uint64)(unsafe.Pointer(blk.offs)), n+1)
		return unsafe.Slice((*byte)(unsafe.Pointer(blk.bytes)), int(ooffs[n])), ooffs, unsafe.Slice((*byte)(unsafe.Pointer(blk.flags)), n), 0, nil
	}, nil
}

// pinnedBlock / pinnedPool: page-locked output blocks (cbh_alloc_pinned) kept between calls.
type pinnedBlock struct {
	base  unsafe.Pointer
	size  int
	cap   int
	bytes *C.uint8_t
	offs  *C.uint64_t
	flags *C.uint8_t
}
type pinnedPool struct {
	mu   sync.Mutex
	free []*pinnedBlock
}

func (p *pinnedPool) get(capBytes, n int) *pinnedBlock {
	want := (capBytes+7)/8*8 + 8*(n+1) + n + 8
	p.mu.Lock()
	for i, b := range p.free {
		if b.size >= want {
			p.free = append(p.free[:i], p.free[i+1:]...)
			p.mu.Unlock()
			b.layout(want-8*(n+1)-n-8, n)
			return b
		}
	}
	p.mu.Unlock()
	base := C.cbh_alloc_pinned(C.size_t(want))
	if base == nil { // hipHostMalloc failed: never lay out - or pool - a block without memory behind it
		return nil
	}
	b := &pinnedBlock{base: base, size: want}
	b.layout(want-8*(n+1)-n-8, n)
	return b
}
func (b *pinnedBlock) layout(capBytes, n int) {
	capBytes = capBytes / 8 * 8
	b.cap = capBytes
	b.bytes = (*C.uint8_t)(b.base)
	b.offs = (*C.uint64_t)(unsafe.Add(b.base, capBytes))
	b.flags = (*C.uint8_t)(unsafe.Add(b.base, capBytes+8*(n+1)))
}

// put returns a block to the pool; the pool keeps at most pinnedPoolMax blocks and frees the rest (page-locked memory is
// not the process's to hoard).
const pinnedPoolMax = 16

func (p *pinnedPool) put(b *pinnedBlock) {
	if b == nil || b.base == nil {
		return
	}
	p.mu.Lock()
	if len(p.free) < pinnedPoolMax {
		p.free = append(p.free, b)
		b = nil
	}
	p.mu.Unlock()
	if b != nil {
		C.cbh_free_pinned(b.base)
	}
}

type traced struct {
	bytes      []byte // serialized CheckOutput holding only outputs (6) and evaluation_errors (7)
	incomplete bool
}

// traceGPU runs cbh_trace_batch over the inputs cbi_table_trace_scope selects and decodes its log with cbi_trace_pb.
func (g *gpuEngine) traceGPU(buf []byte, offs []C.uint64_t, oflags []byte, n int, dv, ds *C.char, params *C.cbh_params) (map[int]traced, error) {
	scope := C.cbi_table_trace_scope(g.ingest)
	if scope == 0 {
		return nil, nil
	}
	var sel []int
	for i := 0; i < n; i++ {
		if oflags[i]&C.CBI_OUT_UNSUPPORTED != 0 {
			continue
		}
		if scope == 2 || oflags[i]&C.CBI_OUT_CEL_ERROR != 0 {
			sel = append(sel, i)
		}
	}
	if len(sel) == 0 {
		return nil, nil
	}
	sbuf := make([]byte, 0, len(buf))
	soffs := make([]C.uint64_t, 1, len(sel)+1)
	for _, i := range sel {
		sbuf = append(sbuf, buf[offs[i]:offs[i+1]]...)
		soffs = append(soffs, C.uint64_t(len(sbuf)))
	}
	if len(sbuf) == 0 {
		sbuf = append(sbuf, 0)[:1]
	}
	var pin runtime.Pinner
	pin.Pin(&sbuf[0])
	pin.Pin(&soffs[0])
	defer pin.Unpin()
	sptr := (*C.uint8_t)(unsafe.Pointer(&sbuf[0]))

	var batch *C.cbi_batch
	if C.cbi_flatten_pb(g.ingest, sptr, &soffs[0], C.uint32_t(len(sel)), dv, ds, 1, &batch) != 0 {
		return nil, errors.New(C.GoString(C.cbi_last_error()))
	}
	defer C.cbi_batch_free(batch)
	view := C.cbi_batch_view(batch)
	nt, nr := C.size_t(view.n_tuples), C.size_t(view.n_requests)
	res := C.cbh_result{
		effect:   (*C.uint8_t)(C.calloc(nt+1, 1)),
		status:   (*C.uint8_t)(C.calloc(nt+1, 1)),
		edr_mask: (*C.uint64_t)(C.calloc(nr+1, 8)),
	}
	defer func() {
		C.free(unsafe.Pointer(res.effect))
		C.free(unsafe.Pointer(res.status))
		C.free(unsafe.Pointer(res.edr_mask))
	}()
	tr := C.cbh_trace{capacity: C.uint32_t(4*nt + 256)}
	for {
		tr.records = (*C.uint32_t)(C.calloc(C.size_t(tr.capacity), 4*C.CBH_TRACE_RECORD_WORDS))
		rc := C.cbh_trace_batch(g.table, view, params, &res, &tr)
		if rc != 0 {
			C.free(unsafe.Pointer(tr.records))
			return nil, errors.New(C.GoString(C.cbh_last_error()))
		}
		if tr.count <= tr.capacity {
			break
		}
		C.free(unsafe.Pointer(tr.records)) // the log overflowed: count says how much room it needs
		tr.capacity = tr.count + 64
	}
	defer C.free(unsafe.Pointer(tr.records))

	var decoded *C.cbi_outputs
	if C.cbi_trace_pb(g.ingest, batch, &res, tr.records, tr.count, sptr, &soffs[0], C.uint32_t(len(sel)), &decoded) != 0 {
		return nil, errors.New(C.GoString(C.cbi_last_error()))
	}
	defer C.cbi_outputs_free(decoded)
	doffs := unsafe.Slice((*uint64)(unsafe.Pointer(C.cbi_outputs_offsets(decoded))), len(sel)+1)
	dflags := unsafe.Slice((*byte)(unsafe.Pointer(C.cbi_outputs_flags(decoded))), len(sel))
	dbytes := unsafe.Slice((*byte)(unsafe.Pointer(C.cbi_outputs_bytes(decoded))), int(doffs[len(sel)]))
	out := make(map[int]traced, len(sel))
	for j, i := range sel {
		out[i] = traced{
			bytes:      append([]byte(nil), dbytes[doffs[j]:doffs[j+1]]...),
			incomplete: dflags[j]&(C.CBI_TRACE_ERRORS_INCOMPLETE|C.CBI_TRACE_OUTPUTS_INCOMPLETE) != 0,
		}
	}
	return out, nil
}

// checkRequestsGPU is the device road for what the server receives: the bytes of MANY CheckResourcesRequests (a coalescer in front
// of svc.CheckResources hands them over as they came off the wire) in, per request the serialized CheckOutputs of its resource
// entries out - cbh_wire_check_requests_pb splits every request into the CheckInputs of cerbos_svc.go:274-288 on the device.
// aux[r] = the serialized engine AuxData cs.auxData.Extract derived for request r, or nil.  outs[r][e] = CheckOutput bytes of resource
// entry e of request r; flags as cbi_outputs_flags (CBI_OUT_*: which entries want the trace pass / the CPU path).  With decision logs
// enabled the call is cbh_wire_check_requests_trail_pb (one more argument: masks [n][(cbh_table_num_policies+31)/32]uint32) and
// masks[r] names the EffectivePolicies of request r's audit entry - see effectivePolicyKeys.
func (g *gpuEngine) checkRequestsGPU(reqs [][]byte, aux [][]byte, p evaluator.EvalParams) (outs [][][]byte, oflags []byte, includeMeta []bool, err error) {
	n := len(reqs)
	offs := make([]C.uint64_t, n+1)
	aoffs := make([]C.uint64_t, n+1)
	total, atotal := 0, 0
	for r := range reqs {
		total += len(reqs[r])
		offs[r+1] = C.uint64_t(total)
		if aux != nil {
			atotal += len(aux[r])
		}
		aoffs[r+1] = C.uint64_t(atotal)
	}
	buf, abuf := make([]byte, 0, total+8), make([]byte, 0, atotal+8)
	for r := range reqs {
		buf = append(buf, reqs[r]...)
		if aux != nil {
			abuf = append(abuf, aux[r]...)
		}
	}
	buf, abuf = append(buf, 0), append(abuf, 0)
	var pin runtime.Pinner
	defer pin.Unpin()
	pin.Pin(&buf[0])
	pin.Pin(&abuf[0])
	var auxPtr *C.uint8_t
	var aoffPtr *C.uint64_t
	if atotal > 0 {
		auxPtr, aoffPtr = (*C.uint8_t)(unsafe.Pointer(&abuf[0])), &aoffs[0]
	}
	dv, ds := C.CString(p.DefaultPolicyVersion), C.CString(p.DefaultScope)
	defer C.free(unsafe.Pointer(dv))
	defer C.free(unsafe.Pointer(ds))
	flags := C.uint32_t(C.CBH_F_WANT_DERIVED_ROLES)
	if p.LenientScopeSearch {
		flags |= C.CBH_F_LENIENT_SCOPE_SEARCH
	}
	if p.StrictEvaluation {
		flags |= C.CBH_F_STRICT_EVALUATION
	}
	params := C.cbh_params{now_ns: C.int64_t(p.NowFunc().UnixNano()), flags: flags}
	first := make([]C.uint32_t, n+1)
	rflags := make([]byte, n+1)
	inputsCap, bytesCap := 8*n+8, 1<<16
	for try := 0; try < 3; try++ {
		out, ooffs, of := make([]byte, bytesCap), make([]C.uint64_t, inputsCap+1), make([]byte, inputsCap+1)
		var need C.size_t
		var info C.cbh_wire_info
		rc := C.cbh_wire_check_requests_pb(g.table, 0, (*C.uint8_t)(unsafe.Pointer(&buf[0])), &offs[0], C.uint32_t(n), auxPtr, aoffPtr, dv, ds, nil, 0, &params,
			&first[0], (*C.uint8_t)(unsafe.Pointer(&rflags[0])), (*C.uint8_t)(unsafe.Pointer(&out[0])), C.size_t(bytesCap), &ooffs[0],
			(*C.uint8_t)(unsafe.Pointer(&of[0])), C.size_t(inputsCap), &need, &info)
		if rc == 2 { // the outputs or the inputs outgrew the buffers: both sizes are exact now
			bytesCap, inputsCap = max(bytesCap, int(need)+64), max(inputsCap, int(info.n_requests))
			continue
		}
		if rc != 0 { // 1: some entry is the host flattener's (the caller takes checkResourcesGPU per request); < 0: info.first_bad names the request
			return nil, nil, nil, errors.New(C.GoString(C.cbh_last_error()))
		}
		outs, includeMeta = make([][][]byte, n), make([]bool, n)
		for r := 0; r < n; r++ {
			includeMeta[r] = rflags[r]&1 != 0
			for i := int(first[r]); i < int(first[r+1]); i++ {
				outs[r] = append(outs[r], out[ooffs[i]:ooffs[i+1]])
			}
		}
		return outs, of[:int(first[n])], includeMeta, nil
	}
	return nil, nil, nil, errors.New("cbh_wire_check_requests_pb: buffers kept growing")
}

// effectivePolicyKeys turns one row of cbh_check_batch_trail's / cbh_wire_check_requests_trail_pb's masks into the keys of
// AuditTrail.EffectivePolicies: the policies whose bit is set and, for a scoped resource or principal policy, those of its ancestors
// the table holds (the source attributes a policy set carries, compile.go:153-180).  keys[i] = cbh_table_policy_key(t, i).
func effectivePolicyKeys(keys []string, mask []uint32) []string {
	have := make(map[string]struct{}, len(keys))
	for _, k := range keys {
		have[k] = struct{}{}
	}
	out := map[string]struct{}{}
	for i, k := range keys {
		if mask[i>>5]>>(uint(i)&31)&1 == 0 {
			continue
		}
		out[k] = struct{}{}
		if slash := strings.IndexByte(k, '/'); slash >= 0 && !strings.HasPrefix(k, "role.") {
			head, scope := k[:slash], k[slash+1:]
			for scope != "" {
				if dot := strings.LastIndexByte(scope, '.'); dot >= 0 {
					scope = scope[:dot]
				} else {
					scope = ""
				}
				anc := head
				if scope != "" {
					anc = head + "/" + scope
				}
				if _, ok := have[anc]; ok {
					out[anc] = struct{}{}
				}
			}
		}
	}
	res := make([]string, 0, len(out))
	for k := range out {
		res = append(res, k)
	}
	sort.Strings(res)
	return res
}

// checkResourcesGPU serves one CheckResourcesRequest without building CheckInputs (svc/cerbos_svc.go:255-344 would
// call this instead of cs.eng.Check when the engine is a GPU engine): the request's own bytes go in, the serialized
// CheckResourcesResponse comes back.  auxData is what cs.auxData.Extract returned for the request's JWT (may be nil).
// fallback[i] is true where resource entry i must be evaluated on the CPU path and patched into the response.
func (g *gpuEngine) checkResourcesGPU(reqBytes []byte, auxData *enginev1.AuxData, p evaluator.EvalParams) (respBytes []byte, fallback []bool, err error) {
	var auxBytes []byte
	if auxData != nil {
		if auxBytes, err = proto.Marshal(auxData); err != nil {
			return nil, nil, err
		}
	}
	if len(reqBytes) == 0 { // an empty request has no resource entries (protovalidate rejects it upstream)
		return nil, nil, errors.New("empty CheckResourcesRequest")
	}
	var pin runtime.Pinner
	pin.Pin(&reqBytes[0])
	defer pin.Unpin()
	reqPtr := (*C.uint8_t)(unsafe.Pointer(&reqBytes[0]))
	var auxPtr *C.uint8_t
	if len(auxBytes) > 0 {
		pin.Pin(&auxBytes[0])
		auxPtr = (*C.uint8_t)(unsafe.Pointer(&auxBytes[0]))
	}
	dv, ds := C.CString(p.DefaultPolicyVersion), C.CString(p.DefaultScope)
	defer C.free(unsafe.Pointer(dv))
	defer C.free(unsafe.Pointer(ds))

	var batch *C.cbi_batch
	if C.cbi_flatten_request_pb(g.ingest, reqPtr, C.uint64_t(len(reqBytes)), auxPtr, C.uint64_t(len(auxBytes)), dv, ds, 1, 1, &batch) != 0 {
		return nil, nil, errors.New(C.GoString(C.cbi_last_error()))
	}
	defer C.cbi_batch_free(batch)
	view := C.cbi_batch_view(batch)
	nt, nr := C.size_t(view.n_tuples), C.size_t(view.n_requests)
	res := C.cbh_result{
		effect: (*C.uint8_t)(C.calloc(nt+1, 1)), policy: (*C.uint32_t)(C.calloc(nt+1, 4)), scope: (*C.uint32_t)(C.calloc(nt+1, 4)),
		status: (*C.uint8_t)(C.calloc(nt+1, 1)), edr_mask: (*C.uint64_t)(C.calloc(nr+1, 8)),
	}
	defer func() {
		for _, ptr := range []unsafe.Pointer{unsafe.Pointer(res.effect), unsafe.Pointer(res.policy), unsafe.Pointer(res.scope), unsafe.Pointer(res.status), unsafe.Pointer(res.edr_mask)} {
			C.free(ptr)
		}
	}()
	flags := C.uint32_t(C.CBH_F_WANT_DERIVED_ROLES)
	if p.LenientScopeSearch {
		flags |= C.CBH_F_LENIENT_SCOPE_SEARCH
	}
	if p.StrictEvaluation {
		flags |= C.CBH_F_STRICT_EVALUATION
	}
	params := C.cbh_params{now_ns: C.int64_t(p.NowFunc().UnixNano()), flags: flags}
	if C.cbh_check_batch(g.table, view, &params, &res) != 0 {
		return nil, nil, errors.New(C.GoString(C.cbh_last_error()))
	}
	// a table with output expressions: the same batch once more through the tracing kernel, its outputs into
	// ResultEntry.outputs (cerbos_svc.go:325-327; the response carries no evaluation errors)
	var traced *C.cbi_outputs
	if C.cbi_table_trace_scope(g.ingest) == 2 {
		tres := C.cbh_result{effect: (*C.uint8_t)(C.calloc(nt+1, 1)), status: (*C.uint8_t)(C.calloc(nt+1, 1))}
		defer C.free(unsafe.Pointer(tres.effect))
		defer C.free(unsafe.Pointer(tres.status))
		tr := C.cbh_trace{capacity: C.uint32_t(4*nt + 256)}
		for {
			tr.records = (*C.uint32_t)(C.calloc(C.size_t(tr.capacity), 4*C.CBH_TRACE_RECORD_WORDS))
			if C.cbh_trace_batch(g.table, view, &params, &tres, &tr) != 0 {
				C.free(unsafe.Pointer(tr.records))
				return nil, nil, errors.New(C.GoString(C.cbh_last_error()))
			}
			if tr.count <= tr.capacity {
				break
			}
			C.free(unsafe.Pointer(tr.records))
			tr.capacity = tr.count + 64
		}
		defer C.free(unsafe.Pointer(tr.records))
		if C.cbi_trace_request_pb(g.ingest, batch, &tres, tr.records, tr.count, reqPtr, C.uint64_t(len(reqBytes)), auxPtr, C.uint64_t(len(auxBytes)), &traced) != 0 {
			return nil, nil, errors.New(C.GoString(C.cbi_last_error()))
		}
		defer C.cbi_outputs_free(traced)
	}
	var assembled *C.cbi_outputs
	if C.cbi_assemble_response_traced_pb(g.ingest, batch, &res, reqPtr, C.uint64_t(len(reqBytes)), dv, traced, &assembled) != 0 {
		return nil, nil, errors.New(C.GoString(C.cbi_last_error()))
	}
	defer C.cbi_outputs_free(assembled)
	ooffs := unsafe.Slice((*uint64)(unsafe.Pointer(C.cbi_outputs_offsets(assembled))), 2)
	respBytes = C.GoBytes(unsafe.Pointer(C.cbi_outputs_bytes(assembled)), C.int(ooffs[1]))
	// one flag byte per resource entry; the entry count is the largest input index of the batch + 1
	nEntries := 0
	inputOf := unsafe.Slice((*uint32)(unsafe.Pointer(C.cbi_batch_request_input(batch))), int(view.n_requests))
	for _, i := range inputOf {
		if int(i)+1 > nEntries {
			nEntries = int(i) + 1
		}
	}
	oflags := unsafe.Slice((*byte)(unsafe.Pointer(C.cbi_outputs_flags(assembled))), nEntries)
	fallback = make([]bool, nEntries)
	for i, f := range oflags {
		// also where the device could not build an entry's outputs (CBI_TRACE_OUTPUTS_INCOMPLETE): the CPU path supplies them
		fallback[i] = f&(C.CBI_OUT_UNSUPPORTED|C.CBI_TRACE_OUTPUTS_INCOMPLETE) != 0
	}
	return respBytes, fallback, nil
}
