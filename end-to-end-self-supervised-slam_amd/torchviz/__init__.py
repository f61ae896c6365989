"""Import-surface stand-in for `torchviz` (online_adaption.py:14, train_depth.py:19: imported, never called)."""

_MSG = "out of scope: SURVEY.md section 2 / 5.1 -- autograd-graph rendering is not part of the MI355X hot path"


def make_dot(*args, **kwargs):
    raise NotImplementedError(_MSG)


def make_dot_from_trace(*args, **kwargs):
    raise NotImplementedError(_MSG)
