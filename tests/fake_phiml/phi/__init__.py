""" TEST DOUBLE of the `phi` package surface phiflow_amd's plug-in patches (see tests/fake_phiml/README.md) """
