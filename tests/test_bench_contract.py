"""bench.py's launch contract, checked without a GPU: `--gpus N` outside torchrun must start the N
ranks itself -- as a fresh torch.distributed.run child on 127.0.0.1, before anything in the parent
touches the GPU -- and pass the driver's flags through; under torchrun (WORLD_SIZE set) it must not
spawn again."""
import os
import sys

import pytest

from conftest import ROOT


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    import importlib
    import bench as b
    return importlib.reload(b)


class _FakeRanks:
    """Stands in for the torch.distributed.run child: records the command, 'prints' the given stdout lines."""
    calls = {}
    lines = []

    def __init__(self, cmd, env=None, **kw):
        _FakeRanks.calls = {"cmd": cmd, "env": env, "kw": kw}
        self.stdout = iter(_FakeRanks.lines)

    def wait(self):
        return 0


def test_gpus_flag_spawns_the_ranks(bench, monkeypatch, capsys):
    _FakeRanks.lines = ["[Gloo] Rank 0 is connected to 7 peer ranks.\n", '{"metric": "m", "value": 1}\n']
    monkeypatch.setattr(bench.subprocess, "Popen", _FakeRanks)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: pytest.fail("the parent must not initialise the GPU"))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    calls = _FakeRanks.calls
    cmd = calls["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert calls["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    assert "TEZIP_BENCH_SINGLE_DEVICE" not in calls["env"]           # 8 GPUs visible: one rank per GPU over RCCL
    out = capsys.readouterr()                                         # stdout carries the bench line and nothing else
    assert out.out == '{"metric": "m", "value": 1}\n' and "[Gloo]" in out.err


def test_fewer_gpus_than_ranks_rehearses_on_one_device_over_gloo(bench, monkeypatch):
    _FakeRanks.lines = []
    monkeypatch.setattr(bench.subprocess, "Popen", _FakeRanks)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("TEZIP_BENCH_BACKEND", raising=False)
    assert bench.spawn_ranks(2, ["--gpus", "2"]) == 0
    calls = _FakeRanks.calls
    assert calls["env"]["TEZIP_BENCH_SINGLE_DEVICE"] == "1" and calls["env"]["TEZIP_BENCH_BACKEND"] == "gloo"


def test_flop_accounting_matches_the_design_numbers(bench):
    from tezip_amd.prednet import PredNetConfig
    cfg = PredNetConfig()
    assert bench.live_flops_per_px0(cfg) == 416754      # executed live work per level-0 pixel and frame
    assert bench.conv16_flops_per_px0(cfg) == 405504    # ... of which k_conv16 (levels >= 1)
    # TZ-PA2: the same convolutions as k_wino executes them: 165,888 same-resolution MACs / 2.25 + 36,864 collapsed-tap MACs
    assert bench.wino_executed_flops_per_px0(cfg) == 2 * (165888 // 9 * 4 + 36864) == 221184


def test_the_line_says_which_frames_per_second_value_is(bench):
    """BASELINE.md 4 / SURVEY.md 8d define frames/s host to host; the task contract defines `value` device resident.
    The line carries both and says which is which (checked on the source: the line itself needs a GPU)."""
    import inspect
    src = inspect.getsource(bench.main)
    assert '"value_definition": "device-resident' in src and '"value_host_to_host":' in src
    assert '"value_host_to_host_pipelined":' in src and "pinned_pipelined_payloads_equal_device_resident" in src   # round 5
    # a 64-multiple A convolution counts like a 48-multiple one (the predicate of tz_prednet.hip plain_nt)
    from tezip_amd.prednet import PredNetConfig
    wide = PredNetConfig(stack_sizes=(3, 64, 128))
    # gates of level 1 (E_1 as Winograd + up(R_2) collapsed), gates of level 2, and A_1 (128 = 2 x 64 columns)
    assert bench.wino_executed_flops_per_px0(wide) == 2 * ((4 * 128 + 4 * 128) * 4 * 64 / 4 + 4 * 256 * 4 * 128 / 16 + 4 * 128 * 128 / 4)
    assert '"configs"' in src and '"cfg5_sweep"' in src      # every BASELINE.json config has a driver-run number


def test_the_line_carries_the_host_side_and_the_steps_own_elementwise_roofline(bench):
    """VERDICT r05 items 2 and 7 (checked on the source: the line itself needs a GPU).  SURVEY.md 8(d): PNG decode and zstd are
    excluded from the GPU number and reported separately -- `host_pipeline`: compress.run / decompress.run wall seconds on the
    cfg3 job as PNG files, stage times, zstd level and thread count; `roofline_encode_tail`: the HBM roofline of the
    elementwise tail the timed step launches, beside the stand-alone `roofline_delta`."""
    import inspect
    main, leg = inspect.getsource(bench.main), inspect.getsource(bench.host_pipeline_leg)
    assert 'extras["host_pipeline"] = host_pipeline_leg(' in main
    for key in ("compress_run_s", "decompress_run_s", "compress_frames_per_s", "decompress_frames_per_s", "compress_stages_s",
                "decompress_stages_s", "zstd_threads", "zstd_level", "png_threads", "zstd9_share_of_compress_run",
                "round_trip_max_abs_error", "output_bytes"):
        assert '"%s"' % key in leg, key
    from tezip_amd import compress, decompress
    csrc, usrc = inspect.getsource(compress), inspect.getsource(decompress)
    for stage in ("list + probe", "first window decoded", "remaining windows decoded + staged", "rollout", "encode (payload resident)",
                  "key_frame.dat + entropy.dat", "zstd-9 key_frame.dat (worker)", "payload fetch + zstd-9 entropy.dat"):
        assert '"%s"' % stage in csrc, stage
    for stage in ("zstd-d entropy.dat + stage to HBM", "zstd-d key_frame.dat + stage to HBM", "rollout (decoder)",
                  "decode tail (frames resident)", "frames fetch + PNG encode"):
        assert '"%s"' % stage in usrc, stage
    assert '"roofline_encode_tail":' in main and '"bytes_per_element": tail_bpe' in main and '"roofline_delta":' in main


def test_stage_log_collects_without_printing(monkeypatch, capsys):
    """compress.STAGE_LOG (what the host_pipeline leg reads): marks and worker-side durations land in the list, nothing is
    printed unless TEZIP_TIMING is set, and nothing is recorded when no list is installed."""
    from tezip_amd import compress
    monkeypatch.delenv("TEZIP_TIMING", raising=False)
    monkeypatch.setattr(compress, "STAGE_LOG", [])
    st = compress._Stages("decompress")
    st.mark("a")
    st.add("b (worker)", 0.25)
    assert [(r, n) for r, n, _ in compress.STAGE_LOG] == [("decompress", "a"), ("decompress", "b (worker)")]
    assert compress.STAGE_LOG[1][2] == 0.25 and compress.STAGE_LOG[0][2] >= 0.0
    monkeypatch.setattr(compress, "STAGE_LOG", None)
    compress._Stages().mark("c")
    out = capsys.readouterr()
    assert out.out == "" and out.err == ""
