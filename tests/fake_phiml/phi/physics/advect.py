""" TEST DOUBLE of phi.physics.advect entry points (advect.py:20-24,156-215): record the call, return the field """
CALLS = []


def euler(*a, **k): raise NotImplementedError
def rk4(*a, **k): raise NotImplementedError


def semi_lagrangian(field, velocity, dt, integrator=euler):
    CALLS.append('semi_lagrangian'); return field


def mac_cormack(field, velocity, dt, correction_strength=1.0, integrator=euler):
    CALLS.append('mac_cormack'); return field


def advect(field, velocity, dt, integrator=euler):
    CALLS.append('advect'); return field
