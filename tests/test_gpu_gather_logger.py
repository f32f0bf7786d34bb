"""dist.GatherLogger on the GPU (one rank: the collective degenerates to a copy, everything else -- the two snapshot buffers,
the side stream, the per-buffer `consumed` events the step stream waits on, the timing events -- is the multi-rank code)."""
import pytest
import torch

from gym_minigrid_amd import dist as mdist

pytestmark = pytest.mark.gpu


def test_double_buffered_logger_keeps_every_snapshot_and_times_itself():
    n = 100_000
    dev = torch.device("cuda", 0)
    lg = mdist.GatherLogger(n, dev, world=1)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    reward = torch.zeros(n, dtype=torch.float32, device=dev)
    seen = []
    for k in range(5):
        done.fill_(k % 2)
        reward.fill_(float(k))
        lg.submit(done, reward)
        done.fill_(7)            # the env overwrites its outputs right away: the snapshot must already be private
        reward.fill_(-1.0)
        gd, gr = lg.wait()
        seen.append((int(gd.sum()), float(gr[0]), float(gr[-1])))
        assert torch.equal(gd, lg.done) and torch.equal(gr, lg.reward)
    torch.cuda.synchronize()
    assert seen == [((k % 2) * n, float(k), float(k)) for k in range(5)]
    st = lg.stats()
    assert st["exchanges"] == 5 and st["buffers"] == 2
    assert st["collective_us_mean"] > 0 and st["collective_us_max"] >= st["collective_us_mean"]
    assert st["step_stream_waits"] == 3               # submits 3..5 came back to a buffer that had been used
    assert st["step_stream_wait_us_total"] >= 0 and st["gather_unfinished_when_buffer_reused"] == 0  # (wait() joined every gather first)


def test_device_identity_names_the_gpu():
    i = mdist.device_identity(0)
    assert i["index"] == 0 and i["name"] and ("uuid" in i or "pci" in i) and i["key"]
