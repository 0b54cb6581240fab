"""Frame helpers on the hot path (mirror of reference amof/atom.py).

``get_neighborlist`` is not mirrored as a Python function: its job (ASE
neighbour search, amof/atom.py:72-87) is done inside the HIP kernels
(amof_cn_count / amof_bad_hist)."""

import numpy as np

from . import data as _data


def get_number_density(atom):
    """number density in Angstrom^-3 (reference amof/atom.py:18-22)"""
    return len(atom) / atom.get_volume()


def get_total_mass(atom):
    return np.sum(atom.get_masses())


def get_density(atom):
    """density in kg/L (reference amof/atom.py:11-16)"""
    conversion_factor = 1.66053906660
    return conversion_factor * get_total_mass(atom) / atom.get_volume()


def select_species_positions(atom, atomic_number):
    """positions of one species (reference amof/atom.py:29-42)"""
    if atomic_number is None:
        return atom.get_positions()
    return atom.get_positions()[atom.get_atomic_numbers() == atomic_number]


def get_atomic_numbers_unique(atom):
    """list of atomic numbers present, in Python ``set`` order
    (reference amof/atom.py:44-46; this order fixes DataFrame column order)"""
    return list(set(atom.get_atomic_numbers()))


def format_cutoff(nb_set_and_cutoff, format='ase', sort_pair=False):
    """``{'Zn-N': 2.5}`` -> ``{(30, 7): 2.5}`` (reference amof/atom.py:48-70)"""
    if format == 'ase':
        cutoff_dict = {}
        for nn_set, cutoff in nb_set_and_cutoff.items():
            xx = tuple(_data.atomic_numbers[i] for i in nn_set.split('-'))
            if sort_pair:
                xx = tuple(sorted(xx))
            cutoff_dict[xx] = cutoff
        return cutoff_dict


def cutoff_matrix(cutoff_dict, kinds):
    """Per-species-pair cutoff matrix for the C ABI.

    ASE applies a dict cutoff symmetrically and later items overwrite earlier
    ones ([3P-memory] of ase.neighborlist.primitive_neighbor_list); pairs that
    are absent are never neighbours (0)."""
    lut = {z: k for k, z in enumerate(kinds)}
    m = np.zeros((len(kinds), len(kinds)), dtype=np.float64)
    for (z1, z2), c in cutoff_dict.items():
        if z1 in lut and z2 in lut:
            m[lut[z1], lut[z2]] = c
            m[lut[z2], lut[z1]] = c
    return m
