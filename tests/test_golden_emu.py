""" golden fixtures replayed on CPU (kernel sources under the fiber emulation -- test infrastructure) """
import golden_cases


def test_golden_smoke_plume(emu_backend):
    golden_cases.run_smoke_plume(emu_backend)


def test_golden_taylor_green(emu_backend):
    golden_cases.run_taylor_green(emu_backend)


def test_golden_cavity_obstacle(emu_backend):
    golden_cases.run_cavity_obstacle(emu_backend)


def test_golden_smoke_plume_mac_cormack_first_steps(emu_backend):
    """ BASELINE configs[0] (128 x 128, mac_cormack smoke): the first 2 of the 50 steps -- the full run is a `-m gpu` test """
    golden_cases.run_smoke_plume_mac_cormack(emu_backend, max_steps=2)


def test_scene_files(emu_backend, tmp_path):
    """ window of the reference's own scene files -> fields -> one step vs the oracle -> write / read round trip """
    golden_cases.run_scene_files(emu_backend, tmp_path)
