"""Path helper used by every ``write_to_file`` / ``from_file``.

Behaviour of reference amof/files/path.py:7-21: result files carry a fixed extension ('.rdf', '.msd', '.bad',
'.cn'); a name that already ends with it is left alone, any other name -- including one with a different
extension, e.g. 'run.v2' -- gets it appended (never substituted)."""

import pathlib


def append_suffix(path, suffix):
    """``pathlib.Path`` of ``path`` ending in ``suffix`` (given with or without its leading dot; '' = unchanged)."""
    wanted = suffix if suffix.startswith('.') or not suffix else '.' + suffix
    target = pathlib.Path(path)
    if target.suffix == wanted:
        return target
    return target.parent / (target.name + wanted)
