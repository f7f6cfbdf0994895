// Package muse -- cgo binding of libmuse_hip.so for aouyang1/go-muse.
//
// WRITTEN BLIND: the build image has no Go toolchain, so this file has never
// been compiled.  It is the reference-side binding a maintainer would add next
// to the existing sources (series.go, group.go, labels.go, results.go,
// scores.go stay as they are); it replaces the bodies of NewBatch / Batch.Run
// (muse_batch.go:23-130) and New / Muse.Run (muse.go:23-92) and deletes
// xcorr.go's hot loop.  Every cgo call copies its inputs (no Go pointer is
// retained by C after the call returns), as the cgo pointer rules require.
//
// Builds at the reference's own language level (go.mod:3, go 1.13): C memory is viewed through the array-pointer
// idiom (f64View ... below), not unsafe.Slice / unsafe.Add (Go 1.17).
// Build: CGO_CFLAGS="-I${REPO}/include" CGO_LDFLAGS="-L${REPO}/go-muse_amd/lib -lmuse_hip"
package muse

/*
#cgo LDFLAGS: -lmuse_hip
#include <stdlib.h>
#include "muse_hip.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"sort"
	"sync"
	"sync/atomic"
	"unsafe"
)

// Views of C memory as Go slices (n <= maxView elements; the array types only bound the index arithmetic, nothing of
// that size is ever allocated).
const maxView = 1 << 30

func f64View(p unsafe.Pointer, n int) []float64 { return (*[maxView]float64)(p)[:n:n] }
func i32View(p unsafe.Pointer, n int) []int32   { return (*[maxView]int32)(p)[:n:n] }
func u8View(p *C.uint8_t, n int) []C.uint8_t    { return (*[maxView]C.uint8_t)(unsafe.Pointer(p))[:n:n] }
func recView(p *C.muse_record, n int) []C.muse_record {
	return (*[maxView]C.muse_record)(unsafe.Pointer(p))[:n:n]
}
func batchView(p **C.muse_batch, n int) []*C.muse_batch {
	return (*[maxView]*C.muse_batch)(unsafe.Pointer(p))[:n:n]
}

// hipError turns a muse_status into a Go error carrying muse_last_error().
func hipError(status C.int) error {
	if status == C.MUSE_OK {
		return nil
	}
	return errors.New(C.GoString(C.muse_last_error()))
}

// engine is one muse_ctx (one GPU).  A process-wide default is created lazily.
type engine struct{ ctx *C.muse_ctx }

var (
	defaultEngine    *engine
	defaultEngineErr error
	defaultEngineOne sync.Once
)

// getEngine creates the process-wide context exactly once, however many goroutines race to the first
// use (muse_test.go:203-214 drives one Muse from many goroutines).  A failed creation is remembered:
// there is no CPU fallback to retry into.
func getEngine() (*engine, error) {
	// SetDevices([]int{k}): ONE engine selected -- every unsharded path (Batch, Muse, xCorr) runs on that device, as the
	// Python and C++ mirrors do with engines[0]; the default context on device 0 is not created for it
	engineSetMu.Lock()
	if len(engineSet) == 1 {
		e := engineSet[0]
		engineSetMu.Unlock()
		return e, nil
	}
	engineSetMu.Unlock()
	defaultEngineOne.Do(func() {
		e := &engine{}
		// a cgo call and the muse_last_error that explains it must run on one OS thread
		runtime.LockOSThread()
		defer runtime.UnlockOSThread()
		if err := hipError(C.muse_ctx_create(0, &e.ctx)); err != nil {
			defaultEngineErr = err
			return
		}
		defaultEngine = e
	})
	return defaultEngine, defaultEngineErr
}

// The device set a Batch shards its Comparison group over (SURVEY 8e: the goroutine fan-out of
// muse_batch.go:99-130 becomes one goroutine per GPU, each driving its own muse_ctx).  Default: device 0 only.
var (
	engineSet   []*engine
	engineSetMu sync.Mutex
)

// SetDevices selects the GPUs every later Batch.Run shards over (one muse_ctx per listed device; a device may be
// listed more than once).  A single id selects that device for the unsharded path.  SetDevices(nil) goes back to the
// single default engine (device 0).  DeviceCount() tells how many
// the process can see; SetDevices(AllDevices()) is the eight-GPU configuration of one node.
func SetDevices(ids []int) error {
	engineSetMu.Lock()
	defer engineSetMu.Unlock()
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var set []*engine
	for _, id := range ids {
		e := &engine{}
		if err := hipError(C.muse_ctx_create(C.int32_t(id), &e.ctx)); err != nil {
			for _, d := range set {
				C.muse_ctx_destroy(d.ctx)
			}
			return err
		}
		set = append(set, e)
	}
	for _, d := range engineSet {
		C.muse_ctx_destroy(d.ctx) // (reference-counted: lives until its last group / batch is freed)
	}
	engineSet = set
	return nil
}

// DeviceCount returns the number of gfx950 devices visible to the process.
func DeviceCount() (int, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var n C.int32_t
	if err := hipError(C.muse_device_count(&n)); err != nil {
		return 0, err
	}
	return int(n), nil
}

// AllDevices lists every visible device once.
func AllDevices() []int {
	n, _ := DeviceCount()
	ids := make([]int, n)
	for i := range ids {
		ids[i] = i
	}
	return ids
}

func shardEngines() []*engine {
	engineSetMu.Lock()
	defer engineSetMu.Unlock()
	if len(engineSet) < 2 {
		return nil
	}
	return append([]*engine(nil), engineSet...)
}

// SetScreening switches the filter-and-refine Run of the default engine (include/muse_hip.h:
// muse_ctx_set_screening; OFF by default -- every series is scored in float64 like the reference --;
// when on, Runs over large groups screen every series in fp32 and re-evaluate in fp64 only the rows that
// can reach the top-N: same Scores).
// minRows > 1 sets the smallest group the path is used for.
func SetScreening(enable bool, minRows int) error {
	e, err := getEngine()
	if err != nil {
		return err
	}
	v := C.int32_t(0)
	if enable {
		v = 1
		if minRows > 1 {
			v = C.int32_t(minRows)
		}
	}
	return hipError(C.muse_ctx_set_screening(e.ctx, v))
}

// appendSeries uploads series (all of length n) to a device group through the library's pinned staging windows
// (muse_group_stage / muse_group_commit): every Series is copied ONCE, straight out of its Go slice into pinned C memory, by a
// few goroutines in pieces of ~256 KB; a piece that completes the packed prefix of the window commits that prefix -- runs of
// >= 4 MB, one host-to-device copy command each (a command costs ~13 us whatever its size) -- so the rows cross PCIe while the
// next pieces are packed.  No Go pointer crosses: the window is C memory, the goroutines write into it through f64View.
func appendSeries(g *C.muse_group, series []*Series, n int) error {
	for first := 0; first < len(series); {
		var win *C.double
		var granted C.int64_t
		if err := hipError(C.muse_group_stage(g, C.int64_t(len(series)-first), &win, &granted)); err != nil {
			return err
		}
		k := int(granted)
		rows := f64View(unsafe.Pointer(win), k*n)
		piece := (256 << 10) / (8 * n)
		if piece < 1 {
			piece = 1
		}
		npieces := (k + piece - 1) / piece
		commitPieces := (4 << 20) / (piece * 8 * n)
		if commitPieces < 1 {
			commitPieces = 1
		}
		workers := runtime.GOMAXPROCS(0)
		if workers > 8 {
			workers = 8
		}
		if workers > npieces {
			workers = npieces
		}
		var (
			mu        sync.Mutex
			done      = make([]bool, npieces)
			watermark = 0
			next      int64 = -1
			firstErr  error
			wg        sync.WaitGroup
		)
		for w := 0; w < workers; w++ {
			wg.Add(1)
			go func() {
				defer wg.Done()
				runtime.LockOSThread() // (muse_last_error is per host thread)
				defer runtime.UnlockOSThread()
				for {
					p := int(atomic.AddInt64(&next, 1))
					if p >= npieces {
						return
					}
					lo, hi := p*piece, (p+1)*piece
					if hi > k {
						hi = k
					}
					for r := lo; r < hi; r++ {
						copy(rows[r*n:(r+1)*n], series[first+r].y)
					}
					mu.Lock()
					done[p] = true
					wm := watermark
					for wm < npieces && done[wm] {
						wm++
					}
					if wm > watermark && (wm == npieces || wm-watermark >= commitPieces) {
						clo, chi := watermark*piece, wm*piece
						if chi > k {
							chi = k
						}
						// (every row is committed even after a failure: the window has to close)
						if err := hipError(C.muse_group_commit(g, C.int64_t(clo), C.int64_t(chi-clo))); err != nil && firstErr == nil {
							firstErr = err
						}
						watermark = wm
					}
					mu.Unlock()
				}
			}()
		}
		wg.Wait()
		if firstErr != nil {
			return firstErr
		}
		first += k
	}
	return nil
}

// deviceGroup mirrors Group's rows on the GPU; rows are appended once.
type deviceGroup struct {
	g        *C.muse_group
	uploaded int
}

// residentRows uploads the series added since the last call (Group.Add order).
// Group gains two fields: order []*Series (insertion order) and dev *deviceGroup.
func (g *Group) residentRows(e *engine) (*C.muse_group, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if g.dev == nil {
		d := &deviceGroup{}
		n := g.n
		if n < 1 {
			n = 1
		}
		if err := hipError(C.muse_group_create(e.ctx, C.int64_t(len(g.order)), C.int32_t(n), &d.g)); err != nil {
			return nil, err
		}
		runtime.SetFinalizer(d, func(d *deviceGroup) { C.muse_group_free(d.g) })
		g.dev = d
	}
	if g.dev.uploaded < len(g.order) {
		if err := appendSeries(g.dev.g, g.order[g.dev.uploaded:], g.n); err != nil {
			return nil, err
		}
		g.dev.uploaded = len(g.order)
	}
	return g.dev.g, nil
}

// groupShard is one contiguous row range of the Group on one device (the shard_bounds rule: equal shares rounded up to
// an even row count -- the fused kernels pack two series per pass).  The cut is made once, at the first sharded Run;
// series added later extend the last shard.  Group gains a third field: shards []*groupShard.
type groupShard struct {
	e                *engine
	g                *C.muse_group
	lo, hi, uploaded int
}

func (g *Group) residentShards(es []*engine) ([]*groupShard, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	same := len(g.shards) == len(es)
	for i := 0; same && i < len(es); i++ {
		same = g.shards[i].e == es[i]
	}
	if !same {
		g.shards = nil // (the old shards' device groups are released by their finalizers)
		M, W := len(g.order), len(es)
		per := (M + W - 1) / W
		per = (per + 1) / 2 * 2
		n := g.n
		if n < 1 {
			n = 1
		}
		for r := 0; r < W; r++ {
			sh := &groupShard{e: es[r]}
			sh.lo = r * per
			if sh.lo > M {
				sh.lo = M
			}
			sh.hi = sh.lo + per
			if sh.hi > M {
				sh.hi = M
			}
			if err := hipError(C.muse_group_create(es[r].ctx, C.int64_t(sh.hi-sh.lo), C.int32_t(n), &sh.g)); err != nil {
				return nil, err
			}
			runtime.SetFinalizer(sh, func(sh *groupShard) { C.muse_group_free(sh.g) })
			g.shards = append(g.shards, sh)
		}
	}
	if len(g.shards) > 0 {
		g.shards[len(g.shards)-1].hi = len(g.order) // series added since the cut
	}
	for _, sh := range g.shards {
		if sh.lo+sh.uploaded < sh.hi {
			if err := appendSeries(sh.g, g.order[sh.lo+sh.uploaded:sh.hi], g.n); err != nil {
				return nil, err
			}
			sh.uploaded = sh.hi - sh.lo
		}
	}
	return g.shards, nil
}

// Batch keeps the exported fields of muse_batch.go:13-19; x and n are gone
// (the reference spectrum lives on the device).
type Batch struct {
	ref         []float64
	Comparison  *Group
	Results     *Results
	Concurrency int
	probe       *C.muse_group // empty group the template batch is bound to
	template    *C.muse_batch // owns the reference spectrum: validated and transformed ONCE, in NewBatch
	batch       *C.muse_batch // the batch over the Comparison group's resident rows: shares the template's spectrum
	batchGroup  *C.muse_group
	shardBatch  []*C.muse_batch // sharded Runs (SetDevices): one device batch per shard
	shardGroup  []*C.muse_group
}

// NewBatch replaces muse_batch.go:23-52: same length check, same
// "Invalid input query" error on a constant reference.
func NewBatch(ref *Series, comp *Group, results *Results, cc int) (*Batch, error) {
	for uid, s := range comp.registry {
		if ref.Length() != s.Length() {
			return nil, fmt.Errorf("%s from comparison group series does not have the same length as the reference", uid)
		}
	}
	if cc < 1 {
		cc = 1
	}
	e, err := getEngine()
	if err != nil {
		return nil, err
	}
	b := &Batch{ref: append([]float64(nil), ref.Values()...), Comparison: comp, Results: results, Concurrency: cc}
	// validate the reference now (sigma == 0 -> error), as the reference does
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if err := hipError(C.muse_group_create(e.ctx, 0, C.int32_t(len(b.ref)), &b.probe)); err != nil {
		return nil, err
	}
	st := C.muse_batch_create(e.ctx, b.probe, (*C.double)(unsafe.Pointer(&b.ref[0])), C.int32_t(len(b.ref)), &b.template)
	if err := hipError(st); err != nil {
		C.muse_group_free(b.probe)
		return nil, fmt.Errorf("Invalid input query, %v", err)
	}
	// the device handles are released with the Batch (muse_batch_free takes no error path)
	runtime.SetFinalizer(b, func(b *Batch) {
		if b.batch != nil {
			C.muse_batch_free(b.batch)
			b.batch = nil
		}
		C.muse_batch_free(b.template)
		C.muse_group_free(b.probe)
		for _, sb := range b.shardBatch {
			if sb != nil {
				C.muse_batch_free(sb)
			}
		}
		b.shardBatch = nil
	})
	return b, nil
}

// Run replaces muse_batch.go:99-130.  groupByLabels semantics are unchanged
// (indexLabelValues, group.go:76-104); the goroutine fan-out becomes one fused
// kernel launch over the resident matrix plus a device-side group-max / top-N.
func (b *Batch) Run(groupByLabels []string) error {
	labelValuesSet := b.Comparison.indexLabelValues(groupByLabels)
	if len(labelValuesSet) == 0 {
		return nil
	}
	// group id of every series, in upload order
	pos := make(map[string]int, len(b.Comparison.order))
	for i, s := range b.Comparison.order {
		pos[s.UID()] = i
	}
	gid := make([]C.int32_t, len(b.Comparison.order))
	gi := 0
	for _, lv := range labelValuesSet {
		for _, uid := range b.Comparison.index[lv.ID(lv.Keys())] {
			gid[pos[uid]] = C.int32_t(gi)
		}
		gi++
	}
	// Up to exactFeedMaxGroups label groups Results is fed the reference's own sequence: ONE Score per label group, in group
	// order, through the unchanged Results.Update (muse_batch.go:124-128) -- the heap's history, and with it the order Fetch
	// returns exactly tied scores in and which of them survives at the TopN boundary, also into a Results that earlier Runs
	// have filled (results.go:55-72), on one device or sharded.  Beyond it (Run(nil) over a million series) the device
	// pre-selects the TopN candidates and 24 B x TopN cross the host.
	G := len(labelValuesSet)
	exact := G <= exactFeedMaxGroups
	if es := shardEngines(); es != nil {
		return b.runSharded(es, gid, G, exact)
	}
	e, err := getEngine()
	if err != nil {
		return err
	}
	dg, err := b.Comparison.residentRows(e)
	if err != nil {
		return err
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if b.batch == nil || b.batchGroup != dg {
		if b.batch != nil {
			C.muse_batch_free(b.batch)
		}
		// (no second transform of the reference: the batch over the resident rows shares the template's spectrum)
		st := C.muse_batch_create_like(b.template, dg, &b.batch)
		if err := hipError(st); err != nil {
			return err
		}
		b.batchGroup = dg
	}
	if exact {
		recs := (*C.muse_record)(C.calloc(C.size_t(G), C.size_t(unsafe.Sizeof(C.muse_record{}))))
		defer C.free(unsafe.Pointer(recs))
		state := (*C.uint8_t)(C.calloc(C.size_t(G), 1))
		defer C.free(unsafe.Pointer(state))
		if err := hipError(C.muse_batch_run_groups(b.batch, &gid[0], C.int32_t(G), 0, 1, recs, state)); err != nil {
			return err
		}
		return b.feedGroupWinners(recs, state, 1, G)
	}
	r := b.Results
	top := r.TopN
	if top < 1 {
		top = 1
	}
	idx := make([]C.int64_t, top)
	lag := make([]C.int32_t, top)
	score := make([]C.double, top)
	var cnt C.int32_t
	var mean C.double
	st := C.muse_batch_run(b.batch, &gid[0], C.int32_t(len(labelValuesSet)), C.int32_t(r.MaxLag), C.int32_t(r.TopN),
		C.double(r.Threshold), C.int32_t(r.SignFilter), 1, &idx[0], &lag[0], &score[0], &cnt, &mean)
	if err := hipError(st); err != nil {
		return err
	}
	// ordered drain (muse_batch.go:124-128): feed Results in group order
	order := make([]int, int(cnt))
	for i := range order {
		order[i] = i
	}
	sort.SliceStable(order, func(a, c int) bool { return gid[idx[order[a]]] < gid[idx[order[c]]] })
	for _, k := range order {
		b.Results.Update(Score{Labels: b.Comparison.order[idx[k]].Labels(), Lag: int(lag[k]), PercentScore: float64(score[k])})
	}
	return nil
}

const exactFeedMaxGroups = 65536

// feedGroupWinners merges the shards' per-group records (muse_merge_group_winners: the first shard with a member decides the NaN
// rule, the maximum by |score| wins, the earlier shard on ties) and pushes one Score per label group through Results.Update in
// group order.  recs / state: nShards x G entries in C memory, shard-major.  Call with the OS thread locked.
func (b *Batch) feedGroupWinners(recs *C.muse_record, state *C.uint8_t, nShards, G int) error {
	win := (*C.muse_record)(C.calloc(C.size_t(G), C.size_t(unsafe.Sizeof(C.muse_record{}))))
	defer C.free(unsafe.Pointer(win))
	wst := (*C.uint8_t)(C.calloc(C.size_t(G), 1))
	defer C.free(unsafe.Pointer(wst))
	if err := hipError(C.muse_merge_group_winners(recs, state, C.int32_t(nShards), C.int32_t(G), win, wst)); err != nil {
		return err
	}
	winv := recView(win, G)
	wstv := u8View(wst, G)
	for g := 0; g < G; g++ {
		if wstv[g] != 1 { // 0: no member; 2: the group's score is NaN, which never passes Results.passed
			continue
		}
		w := winv[g]
		b.Results.Update(Score{Labels: b.Comparison.order[int(w.series)].Labels(), Lag: int(w.lag), PercentScore: float64(w.score)})
	}
	return nil
}

// runSharded is Run over several GPUs (SetDevices).  The Comparison group is cut into one contiguous row range per device
// and every shard is scored at the same time, one goroutine per device.  Label groups that live on ONE shard each
// (always true when every series is its own group): each shard returns its top-N candidates (muse_batch_run_shard,
// 24 B x TopN per device) and muse_merge_records selects.  Label groups that straddle shards: each shard returns its
// winner per group, unfiltered (muse_batch_run_groups), and muse_merge_group_records takes the per-group maximum BEFORE
// Results.passed and the top-N heap -- what SURVEY 8e requires for exactness.
func (b *Batch) runSharded(es []*engine, gid []C.int32_t, G int, exact bool) error {
	shards, err := b.Comparison.residentShards(es)
	if err != nil {
		return err
	}
	W := len(shards)
	if len(b.shardBatch) != W {
		for _, sb := range b.shardBatch {
			if sb != nil {
				C.muse_batch_free(sb)
			}
		}
		b.shardBatch = make([]*C.muse_batch, W)
		b.shardGroup = make([]*C.muse_group, W)
	}
	{
		runtime.LockOSThread()
		for r, sh := range shards {
			if b.shardBatch[r] == nil || b.shardGroup[r] != sh.g {
				if b.shardBatch[r] != nil {
					C.muse_batch_free(b.shardBatch[r])
					b.shardBatch[r] = nil
				}
				st := C.muse_batch_create(sh.e.ctx, sh.g, (*C.double)(unsafe.Pointer(&b.ref[0])), C.int32_t(len(b.ref)), &b.shardBatch[r])
				if err := hipError(st); err != nil {
					runtime.UnlockOSThread()
					return err
				}
				b.shardGroup[r] = sh.g
			}
		}
		runtime.UnlockOSThread()
	}
	// does any label group have members on two shards?  (exact: every shard reports per group anyway)
	straddle := exact
	owner := make([]int, G)
	for i := range owner {
		owner[i] = -1
	}
	for r, sh := range shards {
		if exact {
			break
		}
		for i := sh.lo; i < sh.hi && !straddle; i++ {
			o := owner[gid[i]]
			if o >= 0 && o != r {
				straddle = true
			}
			owner[gid[i]] = r
		}
	}
	res := b.Results
	top := res.TopN
	capN := top
	if capN < 1 {
		capN = 1
	}
	per := capN
	if straddle {
		per = G
	}
	// C memory for what the devices write concurrently (muse_record is a plain 24-byte struct)
	recs := (*C.muse_record)(C.calloc(C.size_t(W*per), C.size_t(unsafe.Sizeof(C.muse_record{}))))
	defer C.free(unsafe.Pointer(recs))
	recv := recView(recs, W*per)
	var state *C.uint8_t
	var statev []C.uint8_t
	if straddle {
		state = (*C.uint8_t)(C.calloc(C.size_t(W*G), 1))
		defer C.free(unsafe.Pointer(state))
		statev = u8View(state, W*G)
	}
	cnt := make([]C.int32_t, W)
	errs := make([]error, W)
	var wg sync.WaitGroup
	for r, sh := range shards {
		if sh.lo == sh.hi {
			continue // an empty shard (fewer rows than devices)
		}
		wg.Add(1)
		go func(r int, sh *groupShard) {
			defer wg.Done()
			runtime.LockOSThread() // the cgo call and the muse_last_error that explains it: one OS thread
			defer runtime.UnlockOSThread()
			g := &gid[sh.lo]
			var st C.int
			if straddle {
				st = C.muse_batch_run_groups(b.shardBatch[r], g, C.int32_t(G), C.int64_t(sh.lo), 1, &recv[r*per],
					&statev[r*G])
			} else {
				st = C.muse_batch_run_shard(b.shardBatch[r], g, C.int32_t(G), C.int64_t(sh.lo), C.int32_t(res.MaxLag),
					C.int32_t(top), C.double(res.Threshold), C.int32_t(res.SignFilter), 1, &recv[r*per], &cnt[r])
			}
			errs[r] = hipError(st)
		}(r, sh)
	}
	wg.Wait()
	for _, e := range errs {
		if e != nil {
			return e
		}
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if exact {
		return b.feedGroupWinners(recs, state, W, G)
	}
	idx := make([]C.int64_t, capN)
	lag := make([]C.int32_t, capN)
	score := make([]C.double, capN)
	var n C.int32_t
	var mean C.double
	if straddle {
		st := C.muse_merge_group_records(recs, state, C.int32_t(W), C.int32_t(G), C.int32_t(res.MaxLag), C.int32_t(top),
			C.double(res.Threshold), C.int32_t(res.SignFilter), &idx[0], &lag[0], &score[0], &n, &mean)
		if err := hipError(st); err != nil {
			return err
		}
	} else {
		// compact the shards' candidate lists (count[r] of capN each) in place, then merge
		k := 0
		for r := 0; r < W; r++ {
			for i := 0; i < int(cnt[r]); i++ {
				recv[k] = recv[r*per+i]
				k++
			}
		}
		st := C.muse_merge_records(recs, C.int64_t(k), C.int32_t(top), &idx[0], &lag[0], &score[0], &n, &mean)
		if err := hipError(st); err != nil {
			return err
		}
	}
	order := make([]int, int(n))
	for i := range order {
		order[i] = i
	}
	sort.SliceStable(order, func(a, c int) bool { return gid[idx[order[a]]] < gid[idx[order[c]]] })
	for _, k := range order {
		b.Results.Update(Score{Labels: b.Comparison.order[idx[k]].Labels(), Lag: int(lag[k]), PercentScore: float64(score[k])})
	}
	return nil
}

// RunMany is Run for several batches that share one Comparison group and the
// same Results settings (README.md:10-13: many references against one set of
// series): the resident rows are read and forward-transformed once for all
// references (muse_batch_run_many).  Not part of the reference's API; batches
// that do not qualify are simply run one after the other.
func RunMany(batches []*Batch, groupByLabels []string) error {
	if len(batches) == 0 {
		return nil
	}
	b0 := batches[0]
	same := true
	for _, b := range batches {
		r, r0 := b.Results, b0.Results
		same = same && b.Comparison == b0.Comparison && r.MaxLag == r0.MaxLag && r.TopN == r0.TopN &&
			r.Threshold == r0.Threshold && r.SignFilter == r0.SignFilter
	}
	if !same {
		for _, b := range batches {
			if err := b.Run(groupByLabels); err != nil {
				return err
			}
		}
		return nil
	}
	labelValuesSet := b0.Comparison.indexLabelValues(groupByLabels)
	if len(labelValuesSet) == 0 {
		return nil
	}
	e, err := getEngine()
	if err != nil {
		return err
	}
	dg, err := b0.Comparison.residentRows(e)
	if err != nil {
		return err
	}
	pos := make(map[string]int, len(b0.Comparison.order))
	for i, s := range b0.Comparison.order {
		pos[s.UID()] = i
	}
	gid := make([]C.int32_t, len(b0.Comparison.order))
	gi := 0
	for _, lv := range labelValuesSet {
		for _, uid := range b0.Comparison.index[lv.ID(lv.Keys())] {
			gid[pos[uid]] = C.int32_t(gi)
		}
		gi++
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	// the handle array lives in C memory: cgo forbids passing Go memory that holds pointers
	R := len(batches)
	hs := (**C.muse_batch)(C.malloc(C.size_t(R) * C.size_t(unsafe.Sizeof((*C.muse_batch)(nil)))))
	defer C.free(unsafe.Pointer(hs))
	hv := batchView(hs, R)
	for i, b := range batches {
		if b.batch == nil || b.batchGroup != dg {
			if b.batch != nil {
				C.muse_batch_free(b.batch)
			}
			st := C.muse_batch_create(e.ctx, dg, (*C.double)(unsafe.Pointer(&b.ref[0])), C.int32_t(len(b.ref)), &b.batch)
			if err := hipError(st); err != nil {
				return err
			}
			b.batchGroup = dg
		}
		hv[i] = b.batch
	}
	r := b0.Results
	top := r.TopN
	if top < 1 {
		top = 1
	}
	idx := make([]C.int64_t, R*top)
	lag := make([]C.int32_t, R*top)
	score := make([]C.double, R*top)
	cnt := make([]C.int32_t, R)
	mean := make([]C.double, R)
	st := C.muse_batch_run_many(hs, C.int32_t(R), &gid[0], C.int32_t(len(labelValuesSet)), C.int32_t(r.MaxLag),
		C.int32_t(r.TopN), C.double(r.Threshold), C.int32_t(r.SignFilter), 1, &idx[0], &lag[0], &score[0], &cnt[0], &mean[0])
	if err := hipError(st); err != nil {
		return err
	}
	for i, b := range batches {
		o := i * r.TopN
		order := make([]int, int(cnt[i]))
		for k := range order {
			order[k] = k
		}
		sort.SliceStable(order, func(a, c int) bool { return gid[idx[o+order[a]]] < gid[idx[o+order[c]]] })
		for _, k := range order {
			b.Results.Update(Score{Labels: b.Comparison.order[idx[o+k]].Labels(), Lag: int(lag[o+k]), PercentScore: float64(score[o+k])})
		}
	}
	return nil
}

// xCorrBatch is xCorr (xcorr.go:102-153) for many independent (x, y) pairs in ONE launch (muse_xcorr_batch; SURVEY 8f-4):
// xs[i] and ys[i] are pair i, every x of one length and every y of one length (each zero-padded on its own, xcorr.go:129-130),
// n raised to max(n, lenx, leny) as in xcorr.go:104-106.  Returns per pair the lag, the signed value and whether the
// reference would have returned (nil, 0, 0).  Unexported like xCorr itself: the package's tests are its callers
// (xcorr_test.go:86-202).  The rows are packed into C memory (no Go pointer crosses); inputs are not mutated (the reference's
// zNormalize mutates x and y in place).
func xCorrBatch(xs, ys [][]float64, n int, normalize bool) (lags []int, mvs []float64, isNil []bool, err error) {
	m := len(xs)
	if m == 0 || len(ys) != m {
		return nil, nil, nil, errors.New("xCorrBatch: xs and ys must hold the same, non-zero number of series")
	}
	lenx, leny := len(xs[0]), len(ys[0])
	for i := 0; i < m; i++ {
		if len(xs[i]) != lenx || len(ys[i]) != leny {
			return nil, nil, nil, errors.New("xCorrBatch: every x (and every y) must have one length")
		}
	}
	e, err := getEngine()
	if err != nil {
		return nil, nil, nil, err
	}
	bx := (*C.double)(C.malloc(C.size_t(m) * C.size_t(lenx) * 8))
	by := (*C.double)(C.malloc(C.size_t(m) * C.size_t(leny) * 8))
	clag := (*C.int32_t)(C.malloc(C.size_t(m) * 4))
	cnil := (*C.int32_t)(C.malloc(C.size_t(m) * 4))
	cmv := (*C.double)(C.malloc(C.size_t(m) * 8))
	defer C.free(unsafe.Pointer(bx))
	defer C.free(unsafe.Pointer(by))
	defer C.free(unsafe.Pointer(clag))
	defer C.free(unsafe.Pointer(cnil))
	defer C.free(unsafe.Pointer(cmv))
	if bx == nil || by == nil || clag == nil || cnil == nil || cmv == nil {
		return nil, nil, nil, errors.New("out of memory staging series for xCorrBatch")
	}
	sx := f64View(unsafe.Pointer(bx), m*lenx)
	sy := f64View(unsafe.Pointer(by), m*leny)
	for i := 0; i < m; i++ {
		copy(sx[i*lenx:(i+1)*lenx], xs[i])
		copy(sy[i*leny:(i+1)*leny], ys[i])
	}
	norm := C.int32_t(0)
	if normalize {
		norm = 1
	}
	if err := hipError(C.muse_xcorr_batch(e.ctx, bx, by, C.int64_t(m), C.int32_t(lenx), C.int32_t(leny), C.int32_t(n), norm,
		clag, cmv, cnil, nil)); err != nil {
		return nil, nil, nil, err
	}
	lags, mvs, isNil = make([]int, m), make([]float64, m), make([]bool, m)
	gl := i32View(unsafe.Pointer(clag), m)
	gn := i32View(unsafe.Pointer(cnil), m)
	gv := f64View(unsafe.Pointer(cmv), m)
	for i := 0; i < m; i++ {
		lags[i], mvs[i], isNil[i] = int(gl[i]), gv[i], gn[i] != 0
	}
	return lags, mvs, isNil, nil
}

// Muse keeps the exported fields of muse.go:13-17; x and n are gone (the
// reference spectrum lives on the device, owned by the template batch).
type Muse struct {
	Results  *Results
	ref      []float64
	probe    *C.muse_group // empty group the template batch is bound to
	template *C.muse_batch // owns the reference spectrum, shared by every Run
}

// New replaces muse.go:23-42: empty reference -> error, sigma(ref) == 0 ->
// "Invalid input query" error; the spectrum is computed once, here.
func New(ref *Series, results *Results) (*Muse, error) {
	if ref.Length() < 1 {
		return nil, errors.New("Reference series length must be greater than zero")
	}
	e, err := getEngine()
	if err != nil {
		return nil, err
	}
	m := &Muse{Results: results, ref: append([]float64(nil), ref.Values()...)}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if err := hipError(C.muse_group_create(e.ctx, 0, C.int32_t(len(m.ref)), &m.probe)); err != nil {
		return nil, err
	}
	st := C.muse_batch_create(e.ctx, m.probe, (*C.double)(unsafe.Pointer(&m.ref[0])), C.int32_t(len(m.ref)), &m.template)
	if err := hipError(st); err != nil {
		C.muse_group_free(m.probe)
		return nil, fmt.Errorf("Invalid input query, %v", err)
	}
	runtime.SetFinalizer(m, func(m *Muse) {
		C.muse_batch_free(m.template)
		C.muse_group_free(m.probe)
	})
	return m, nil
}

// Run replaces muse.go:46-92: one label group per call, signed score clamped
// to [-1, 1], the group's best |score| goes to Results.  Safe to call from many
// goroutines (muse_test.go:203-214): every call owns its group and batch, the
// spectrum is shared read-only, Results has its mutex.
func (m *Muse) Run(compGraphs []*Series) error {
	if len(compGraphs) == 0 {
		return nil
	}
	N := len(m.ref)
	rows := make([]float64, 0, len(compGraphs)*N)
	for _, s := range compGraphs {
		if s.Length() != N { // muse.go:68-70
			return fmt.Errorf("Encountered a comparison graph with differing length than the reference, %v", s.Labels())
		}
		rows = append(rows, s.Values()...)
	}
	if _, err := getEngine(); err != nil {
		return err
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	// one ABI call: upload, fused kernel, group maximum, record back; signed scores (muse.go:72-76).  The rows are
	// copied before the call returns (cgo rule); Results.Update applies passed() as the reference does.
	var win C.muse_record
	var state C.uint8_t
	st := C.muse_batch_run_rows(m.template, (*C.double)(unsafe.Pointer(&rows[0])), C.int64_t(len(compGraphs)), C.int64_t(N), 0, &win, &state)
	if err := hipError(st); err != nil {
		return err
	}
	if state == 1 && win.series >= 0 {
		m.Results.Update(Score{Labels: compGraphs[win.series].Labels(), Lag: int(win.lag), PercentScore: float64(win.score)})
	}
	return nil
}
