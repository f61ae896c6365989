import argparse


def arguments():
    """reference: utils/arguments.py:4-11."""
    ap = argparse.ArgumentParser(description="MI355X online depth refinement")
    ap.add_argument("--config_path", type=str, required=True, help="path to the YAML configuration")
    ap.add_argument("--name", type=str, default=None, help="name of the run")
    return vars(ap.parse_args())
