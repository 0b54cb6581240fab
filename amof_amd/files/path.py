"""Path helper used by every ``write_to_file`` / ``from_file``
(mirror of reference amof/files/path.py:7-21)."""

import pathlib


def append_suffix(path, suffix):
    """Append ``suffix`` to ``path`` unless it already is its last suffix.

    Args:
        path: pathlib.Path or str
        suffix: str; a leading '.' is added when missing
    Returns:
        pathlib.Path
    """
    if len(suffix) != 0 and suffix[0] != '.':
        suffix = '.' + suffix
    path = pathlib.Path(path)
    if path.suffix != suffix:
        path = path.parent / (path.name + suffix)
    return path
