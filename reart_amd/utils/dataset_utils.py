"""Host-side helper mirrored from the reference's ``utils/dataset_utils.py`` (hot-path subset)."""
import pickle


def load_normalize_dict(normalize_file):
    """category -> {'centroid', 'scale'} (the reference pickles it next to the data; run_robot.py:72-75)."""
    with open(normalize_file, "rb") as f:
        return pickle.load(f)
