""" golden fixtures replayed on CPU (kernel sources under the fiber emulation -- test infrastructure) """
import golden_cases


def test_golden_smoke_plume(emu_backend):
    golden_cases.run_smoke_plume(emu_backend)


def test_golden_taylor_green(emu_backend):
    golden_cases.run_taylor_green(emu_backend)


def test_golden_cavity_obstacle(emu_backend):
    golden_cases.run_cavity_obstacle(emu_backend)
