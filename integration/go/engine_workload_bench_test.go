// Go CPU baseline for the synthetic workloads of BASELINE.json (C1..C5), to be run inside the reference tree:
//
//	python tools/export_workload.py C2 /tmp/c2            # in this repository
//	go test ./internal/engine/ -run '^$' -bench Workload -workload /tmp/c2 -batch 1000
//
// Reports decisions/s of evaluator.Check over the exported inputs, with GOMAXPROCS as given (run once with
// -cpu 1 for the per-core figure).  The numbers belong next to bench.py's `cpu_baseline` (kind "reference").
package engine_test

import (
	"bufio"
	"context"
	"flag"
	"os"
	"path/filepath"
	"runtime"
	"testing"

	"github.com/stretchr/testify/require"
	"google.golang.org/protobuf/encoding/protojson"

	enginev1 "github.com/cerbos/cerbos/api/genpb/cerbos/engine/v1"
	"github.com/cerbos/cerbos/internal/compile"
	"github.com/cerbos/cerbos/internal/evaluator"
	"github.com/cerbos/cerbos/internal/ruletable"
	"github.com/cerbos/cerbos/internal/schema"
	"github.com/cerbos/cerbos/internal/storage/disk"
)

var (
	workloadDir = flag.String("workload", "", "directory written by tools/export_workload.py")
	batchSize   = flag.Int("batch", 1000, "CheckInputs per Check call")
)

func loadWorkloadInputs(tb testing.TB, path string) []*enginev1.CheckInput {
	tb.Helper()
	f, err := os.Open(path)
	require.NoError(tb, err)
	defer f.Close()

	var inputs []*enginev1.CheckInput
	sc := bufio.NewScanner(f)
	sc.Buffer(make([]byte, 1<<20), 1<<26)
	for sc.Scan() {
		in := &enginev1.CheckInput{}
		require.NoError(tb, protojson.Unmarshal(sc.Bytes(), in))
		inputs = append(inputs, in)
	}
	require.NoError(tb, sc.Err())
	return inputs
}

func workloadEvaluator(tb testing.TB, ctx context.Context, policyDir string) evaluator.Evaluator {
	tb.Helper()
	store, err := disk.NewStore(ctx, &disk.Conf{Directory: policyDir})
	require.NoError(tb, err)
	mgr, err := compile.NewManager(ctx, store)
	require.NoError(tb, err)
	protoRT := ruletable.NewProtoRuletable()
	require.NoError(tb, ruletable.LoadPolicies(ctx, protoRT, mgr))
	require.NoError(tb, ruletable.LoadSchemas(ctx, protoRT, store))
	rt, err := ruletable.NewRuleTable(protoRT)
	require.NoError(tb, err)
	conf := &evaluator.Conf{}
	conf.SetDefaults()
	eval, err := rt.Evaluator(conf, schema.NewConf(schema.EnforcementNone))
	require.NoError(tb, err)
	return eval
}

func BenchmarkWorkload(b *testing.B) {
	if *workloadDir == "" {
		b.Skip("-workload not given")
	}
	ctx, cancel := context.WithCancel(b.Context())
	b.Cleanup(cancel)
	eval := workloadEvaluator(b, ctx, filepath.Join(*workloadDir, "policies"))
	inputs := loadWorkloadInputs(b, filepath.Join(*workloadDir, "inputs.jsonl"))
	require.NotEmpty(b, inputs)

	decisions := 0
	for _, in := range inputs {
		decisions += len(in.GetActions())
	}
	b.Logf("%d inputs, %d decisions per pass, GOMAXPROCS=%d", len(inputs), decisions, runtime.GOMAXPROCS(0))

	b.ReportAllocs()
	b.ResetTimer()
	for b.Loop() {
		for lo := 0; lo < len(inputs); lo += *batchSize {
			hi := min(lo+*batchSize, len(inputs))
			out, err := eval.Check(ctx, inputs[lo:hi])
			if err != nil {
				b.Fatal(err)
			}
			if len(out) != hi-lo {
				b.Fatalf("got %d outputs for %d inputs", len(out), hi-lo)
			}
		}
	}
	b.StopTimer()
	b.ReportMetric(float64(decisions)*float64(b.N)/b.Elapsed().Seconds(), "decisions/s")
}
