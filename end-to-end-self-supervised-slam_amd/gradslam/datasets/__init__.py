"""ICL / TUM loaders are the N2 scope row: they need PNG decoding and the datasets themselves,
neither of which exists on the build or GPU machines.  The synthetic sequence generator the tests and bench
use lives in e2ehip.synthetic."""


class _Unavailable:
    def __init__(self, *a, **k):
        raise NotImplementedError("gradslam.datasets.ICL/TUM loaders are not built yet (SURVEY.md 8f row N2); "
                                  "use e2ehip.synthetic.make_sequence for a synthetic RGB-D sequence")


ICL = TUM = _Unavailable
