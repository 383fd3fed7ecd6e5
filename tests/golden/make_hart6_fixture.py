"""Generates tests/golden/hart6_reference.npz by RUNNING the reference's own objective code.

    python tests/golden/make_hart6_fixture.py          (build container only: needs /root/reference)

benchmark_functions.py is the one reference file that imports with numpy alone.  This script
imports it from the reference checkout, evaluates `hart6().f` and `scale_function(hart6())` (f,
restore, scale) at 64 seeded points and stores inputs and outputs.  The fixture holds data only;
the reference file itself never travels.  It pins the objective (training targets of every GP
fixture here) -- not the acquisition path, whose reference solvers cannot be run.
"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('OEI_REFERENCE_ROOT', '/root/reference')


def main():
    spec = importlib.util.spec_from_file_location('ref_benchmark_functions',
                                                  os.path.join(REF, 'benchmark_functions.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rs = np.random.RandomState(2024)
    X_unit = rs.uniform(0.0, 1.0, (64, 6))
    X_unit[0] = ref.hart6().min                      # the published minimiser
    X_scaled = rs.uniform(-0.5, 0.5, (64, 6))
    h = ref.hart6()
    sf = ref.scale_function(ref.hart6())
    np.savez_compressed(os.path.join(HERE, 'hart6_reference.npz'),
                        X_unit=X_unit, f_unit=h.f(X_unit),
                        X_scaled=X_scaled, f_scaled=sf.f(X_scaled),
                        restored=sf.restore(X_scaled), rescaled=sf.scale(X_unit),
                        bounds=sf.bounds, fmin=np.array(h.fmin))
    print('hart6 at the published minimiser:', float(h.f(X_unit[:1])[0, 0]))


if __name__ == '__main__':
    main()
