"""CPU: pcr_amd/lazylog.py -- the log dict whose entries stay on the device until read, and (round 6) whose rank-averaged
entries can be DEFERRED so that a trainer averages them in the tail of its gradient bucket (one collective per iteration,
none inside a captured forward + backward).  Reference behaviour restated: mmdet's BaseDetector._parse_losses (call site
mmdet3d/models/ReIDNet.py:728) -- every logged scalar averaged over the ranks."""
from collections import OrderedDict

import torch

from pcr_amd import lazylog
from pcr_amd.lazylog import LazyScalars


def test_entries_are_read_lazily_and_keep_their_order_and_types():
    lv = LazyScalars()
    lv.add_device(["match_loss", "num_preds_1"], torch.tensor([0.25, 3.0]), ints=[False, True])
    lv["plain"] = 7
    assert list(OrderedDict.keys(lv)) == ["match_loss", "num_preds_1", "plain"]
    assert lv["match_loss"] == 0.25 and lv["num_preds_1"] == 3 and isinstance(lv["num_preds_1"], int)
    assert dict(lv.items())["plain"] == 7


def test_deferred_entries_wait_for_their_average():
    lv = LazyScalars()
    mine = torch.tensor([1.0, 2.0])
    lv.add_device(["reid_loss", "loss"], mine, reduce=True)
    assert [n for n, _, _ in lv.deferred()] == [["reid_loss", "loss"]] and not lv._pending
    other = LazyScalars()
    other.add_device(["match_acc"], torch.tensor([0.5]))
    lv.update(other)                                  # (train_step merges the model's own entries in)
    assert len(lv.deferred()) == 1
    lv.resolve([torch.tensor([1.5, 2.5])])            # what the bucket's tail came back with
    assert not lv.deferred()
    assert lv["reid_loss"] == 1.5 and lv["loss"] == 2.5 and lv["match_acc"] == 0.5


def test_unresolved_deferred_entries_fall_back_to_this_ranks_values():
    lv = LazyScalars()
    lv.add_device(["loss"], torch.tensor([4.0]), reduce=True)
    assert lv["loss"] == 4.0 and not lv.deferred()


def test_defer_flag_is_scoped():
    assert lazylog.DEFER_REDUCE is False
    with lazylog.defer_reduce(True):
        assert lazylog.DEFER_REDUCE is True
        with lazylog.defer_reduce(False):
            assert lazylog.DEFER_REDUCE is False
        assert lazylog.DEFER_REDUCE is True
    assert lazylog.DEFER_REDUCE is False


def test_parse_losses_defers_only_under_the_flag_and_a_process_group():
    """without torch.distributed initialised nothing is deferred, flag or not (one process: nothing to average)"""
    from mmdet3d.models.ReIDNet import ReIDNet
    losses = {"reid_loss": torch.tensor(0.75)}
    with lazylog.defer_reduce(True):
        loss, lv = ReIDNet._parse_losses(losses)
    assert float(loss) == 0.75 and not lv.deferred() and lv["loss"] == 0.75 and lv["reid_loss"] == 0.75
