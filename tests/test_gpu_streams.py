"""CU-masked streams (yl_stream_create_masked) and the per-unit census probe (yl_xcc_census) on the GPU."""
import pytest
import torch

from yololite_amd import _lib

pytestmark = pytest.mark.gpu


def test_cu_slice_stream_confines_work_to_a_slice_of_every_xcd(gpu_device):
    full = _lib.xcc_census(torch.cuda.current_stream(gpu_device), 8192)
    assert int((full > 0).sum()) == _lib.NUM_XCD * _lib.CUS_PER_XCD
    half = _lib.xcc_census(_lib.cu_slice_stream(gpu_device, 0, 2), 8192)
    assert int(half.sum()) == 8192
    assert int((half > 0).sum()) == _lib.NUM_XCD * _lib.CUS_PER_XCD // 2
    # the dispatcher still uses all eight XCDs: a stream gets a slice of each, never an XCD of its own
    assert all(int(v) > 0 for v in half.sum((1, 2)))
    other = _lib.xcc_census(_lib.cu_slice_stream(gpu_device, 1, 2), 8192)
    assert int(((half > 0) & (other > 0)).sum()) == 0


def test_hot_path_on_a_masked_stream_is_bit_identical(gpu_device):
    """The engine takes whatever stream is current: a CU-masked one changes where the work runs, not the result."""
    from yololite_amd.nn.synth import synth_state_dict
    from yololite_amd.nn import arch
    from yololite_amd.nn.engine import HipEngine

    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    sd = synth_state_dict("n", 8, cls_bias=[-2.0, -2.0, -2.0], cls_gain=4.0)
    eng = HipEngine.from_model(specs, sd, 2, 64, 96, gpu_device)
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(5)).to(gpu_device)
    ref, cref = eng.detect(x, 0.25, 0.7, max_det=50)
    ref, cref = ref.clone(), cref.clone()
    st = _lib.cu_slice_stream(gpu_device, 1, 4)
    st.wait_stream(torch.cuda.current_stream(gpu_device))
    with torch.cuda.stream(st):
        out, cnt = eng.detect(x, 0.25, 0.7, max_det=50, slot=1)
    st.synchronize()
    assert torch.equal(cnt, cref)
    for i, n in enumerate(cnt.tolist()):  # rows past the count are padding (not written by the eager form)
        assert n > 0 and torch.equal(out[i, :n], ref[i, :n])


def test_predict_stream_pipelines_batches_bit_identically(gpu_device, monkeypatch):
    """predict(stream=True) on a source of several batches keeps YL_PREDICT_INFLIGHT engine slots in flight (reference
    generator: predictor.py:149-229, one batch at a time).  Every image's boxes must equal what predict() returns for
    its batch alone, bit for bit, in source order, with the right paths."""
    from yololite_amd import YOLOLite
    from yololite_amd.data.loaders import LoadTensorBatches

    monkeypatch.setenv("YL_PREDICT_INFLIGHT", "3")
    g = torch.Generator().manual_seed(5)
    batches = [torch.randint(0, 256, (2, 3, 96, 128), generator=g, dtype=torch.uint8).to(gpu_device) for _ in range(7)]
    model = YOLOLite("yolo11n.yaml")
    kw = dict(conf=0.001, iou=0.7, device=str(gpu_device), save=False, verbose=False)
    alone = [model.predict(b, **kw) for b in batches]
    gen = model.predict(LoadTensorBatches(batches), stream=True, **kw)
    got = list(gen)
    assert len(got) == 14
    k = 0
    for rs in alone:
        for r in rs:
            assert torch.equal(got[k].boxes.data, r.boxes.data), k
            assert got[k].path == f"image{k}.jpg"
            k += 1
    assert sum(len(r.boxes) for r in got) > 0
    # and a second pass through the same slots / graphs with other inputs
    again = list(model.predict(LoadTensorBatches(batches[::-1]), stream=True, **kw))
    for j, rs in enumerate(alone[::-1]):
        for i, r in enumerate(rs):
            assert torch.equal(again[2 * j + i].boxes.data, r.boxes.data)


def test_native_multi_step_launch_equals_detect(gpu_device):
    """yl_engine_detect_graph_prepare + yl_graph_launch_many (one native call issuing the prepared steps of several slots on
    their streams, what bench.py's timed loop uses): every slot's records equal a plain eager detect on that slot's input,
    also after the inputs changed in place and over two rounds."""
    from yololite_amd.nn import arch
    from yololite_amd.nn.engine import HipEngine
    from yololite_amd.nn.synth import synth_state_dict

    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    sd = synth_state_dict("n", 3, cls_bias=[-3.0, -3.0, -3.0], cls_gain=6.0)
    eng = HipEngine.from_model(specs, sd, 2, 96, 128, gpu_device)
    g = torch.Generator().manual_seed(21)
    S = 3
    xs = [torch.randint(0, 256, (2, 3, 96, 128), generator=g, dtype=torch.uint8).to(gpu_device) for _ in range(S)]
    streams = [torch.cuda.Stream(device=gpu_device) for _ in range(S)]
    outs, handles = [], []
    for j in range(S):
        with torch.cuda.stream(streams[j]):
            o, c, h = eng.prepare_detect(xs[j], 0.25, 0.7, slot=j)
        outs.append((o, c))
        handles.append(h)
    torch.cuda.synchronize()
    for rnd in range(2):
        for j in range(S):  # new contents in the SAME input tensors (the graphs baked their addresses)
            xs[j].copy_(torch.randint(0, 256, (2, 3, 96, 128), generator=g, dtype=torch.uint8))
        torch.cuda.synchronize()
        eng.launch_many([handles[k % S] for k in range(2 * S)], [streams[k % S] for k in range(2 * S)])  # two steps per slot
        torch.cuda.synchronize()
        total = 0
        for j in range(S):
            ro, rc = eng.detect(xs[j], 0.25, 0.7, graph=False, slot=j)
            torch.cuda.synchronize()
            assert torch.equal(outs[j][1], rc), (rnd, j)
            for i in range(2):
                n = int(rc[i])
                total += n
                assert torch.equal(outs[j][0][i, :n], ro[i, :n]), (rnd, j, i)
        assert total > 0


def test_prepared_graphs_survive_eager_work_of_another_engine(gpu_device):
    """Regression (round 4): the hot path's graphs used to clear their candidate counters with a captured hipMemsetAsync; after
    EAGER engine work elsewhere in the process -- here a second engine, created after the graphs, running detect(graph=False) on the
    default stream -- the replays stopped clearing them: counts grew from round to round, or the process died with a GPU memory
    fault.  The counters are cleared by a kernel now (csrc/nms.hip zero_i32): every replay must equal the other engine's answer."""
    from yololite_amd.nn import arch
    from yololite_amd.nn.engine import HipEngine
    from yololite_amd.nn.synth import synth_state_dict

    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    sd = synth_state_dict("n", 3, cls_bias=[-3.0, -3.0, -3.0], cls_gain=6.0)
    eng = HipEngine.from_model(specs, sd, 2, 96, 128, gpu_device)
    g = torch.Generator().manual_seed(21)
    S = 3
    xs = [torch.randint(0, 256, (2, 3, 96, 128), generator=g, dtype=torch.uint8).to(gpu_device) for _ in range(S)]
    streams = [torch.cuda.Stream(device=gpu_device) for _ in range(S)]
    outs, handles = [], []
    for j in range(S):
        with torch.cuda.stream(streams[j]):
            o, c, h = eng.prepare_detect(xs[j], 0.25, 0.7, slot=j)
        outs.append((o, c))
        handles.append(h)
    torch.cuda.synchronize()
    other = HipEngine.from_model(specs, sd, 2, 96, 128, gpu_device)  # created AFTER the graphs
    for rnd in range(3):
        for j in range(S):
            xs[j].copy_(torch.randint(0, 256, (2, 3, 96, 128), generator=g, dtype=torch.uint8))
        torch.cuda.synchronize()
        eng.launch_many([handles[k % S] for k in range(2 * S)], [streams[k % S] for k in range(2 * S)])
        torch.cuda.synchronize()
        for j in range(S):
            ro, rc = other.detect(xs[j], 0.25, 0.7, graph=False, slot=0)  # eager, on the default stream
            torch.cuda.synchronize()
            assert torch.equal(outs[j][1], rc), (rnd, j, outs[j][1].tolist(), rc.tolist())
            for i in range(2):
                n = int(rc[i])
                assert torch.equal(outs[j][0][i, :n], ro[i, :n]), (rnd, j, i)


@pytest.mark.parametrize("batch", [2, 17])
def test_counts_in_pinned_host_memory_equal_the_device_counts(gpu_device, batch):
    """detect(count_host=True): the last kernel of the graph writes the per-image counts straight into pinned host memory (what
    YOLOLite.predict reads after its one synchronisation) -- equal to the device counts of the plain call, over replays with new
    inputs, with the cached scale table of the predictor (scale_static); batch 17 takes the one-launch NMS, batch 2 the four launches."""
    from yololite_amd.nn import arch
    from yololite_amd.nn.engine import HipEngine
    from yololite_amd.nn.synth import synth_state_dict

    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    sd = synth_state_dict("n", 5, cls_bias=[-3.0, -3.0, -3.0], cls_gain=6.0)
    eng = HipEngine.from_model(specs, sd, batch, 96, 160, gpu_device)
    g = torch.Generator().manual_seed(batch)
    scale = torch.tensor([[0.0, 0.0, 1.0, 160.0, 96.0]] * batch, dtype=torch.float32, device=gpu_device)
    x = torch.randint(0, 256, (batch, 3, 96, 160), generator=g, dtype=torch.uint8).to(gpu_device)
    for rnd in range(3):
        x.copy_(torch.randint(0, 256, (batch, 3, 96, 160), generator=g, dtype=torch.uint8))
        o_ref, c_ref = eng.detect(x, 0.25, 0.7, scale=scale, graph=False)
        o_ref, c_ref = o_ref.clone(), c_ref.cpu()
        out, cnt = eng.detect(x, 0.25, 0.7, scale=scale, graph=True, stage="rebind", count_host=True, scale_static=True)
        assert not cnt.is_cuda and cnt.is_pinned()
        torch.cuda.current_stream(gpu_device).synchronize()
        assert torch.equal(cnt, c_ref), rnd
        assert all(torch.equal(out[i, :n], o_ref[i, :n]) for i, n in enumerate(c_ref.tolist())) and int(c_ref.sum()) > 0
