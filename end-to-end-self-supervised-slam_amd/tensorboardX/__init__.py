"""Import-surface stand-in for `tensorboardX` (train_depth.py:17 imports SummaryWriter and constructs one
unconditionally at :48; it is only written to under VIZ.tensorboard, which is OUT OF SCOPE: SURVEY.md section 5.5)."""

_MSG = "out of scope: SURVEY.md section 2 / 5.5 -- TensorBoard logging (VIZ.tensorboard) is not part of the MI355X hot path"


class SummaryWriter:
    def __init__(self, *args, **kwargs):            # constructing the writer must succeed (train_depth.py:48)
        self.args = args

    def __getattr__(self, name):
        if name.startswith("add_") or name in ("flush", "export_scalars_to_json"):
            def _raise(*a, **k):
                raise NotImplementedError(_MSG)
            return _raise
        raise AttributeError(name)

    def close(self):
        pass
