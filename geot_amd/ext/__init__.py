"""Extension-module-shaped front ends over the C ABI.

Each submodule exposes exactly the callables of one of the reference's pybind11
extension modules, with the same argument order and ownership conventions
(SURVEY.md section 8b), so the reference's Python wrappers can import them
unchanged via ``geot_amd.aliases.install()``:

  pointnet2_ext         <-> pointnet2._ext            (pointnet2/_ext_src/src/bindings.cpp:9-22)
  pointops_cuda         <-> pointops_cuda             (pointops/src/pointops_api.cpp:8-12 and
                                                       openpoints/cpp/pointops/src/pointops_api.cpp:13-25)
  pointnet2_batch_cuda  <-> pointnet2_batch_cuda      (openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24)
"""
