"""Trajectory helpers on the hot path (mirror of reference amof/trajectory.py).

Only ``construct_step`` (amof/trajectory.py:244-283) is host logic here;
``get_delta_pos`` (amof/trajectory.py:285-303) runs inside the MSD kernels."""

import logging

import numpy as np

logger = logging.getLogger(__name__)


def construct_step(**kwargs):
    """Construct the ``Step`` column from various constructors.

    Args:
        delta_Step: int, number of simulation steps between two frames
        first_frame: int, first step
        last_frame: int, last step
        number_of_frames: int
        step: slice object or array
    Return:
        numpy array of steps (None when the arguments do not determine one,
        like the reference)
    """
    delta_Step = kwargs.get('delta_Step', None)
    first_frame = kwargs.get('first_frame', None)
    last_frame = kwargs.get('last_frame', None)
    number_of_frames = kwargs.get('number_of_frames', None)
    step = kwargs.get('step', None)
    try:
        if step is not None:
            if isinstance(step, slice):
                return np.array(list(range(step.start or 0, step.stop, step.step or 1)))
            return np.array(step)
        elif delta_Step is not None:
            if first_frame is not None and last_frame is not None:
                return np.arange(first_frame, last_frame, delta_Step)
            elif number_of_frames is not None:
                if first_frame is None and last_frame is not None:
                    first_frame = last_frame - number_of_frames * delta_Step
                if first_frame is not None:
                    return np.arange(first_frame, first_frame + number_of_frames * delta_Step, delta_Step)
        elif number_of_frames is not None:
            if first_frame is not None and last_frame is not None:
                return np.linspace(first_frame, last_frame, number_of_frames)
    except Exception:
        logger.exception("Cannot construct step from provided args")
        raise ValueError
