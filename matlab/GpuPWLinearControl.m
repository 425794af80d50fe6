% GpuPWLinearControl.m
% Shim for libocs.so (include/ocs.h); subclasses / replaces the reference's Control/PWLinearControl.m (Control/Control.m:4-14).
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_*.py).  See INTEGRATION.md.
classdef GpuPWLinearControl < Control            % kind 1; PWConstant = 2, Chebyshev = 3
   properties
      nControlPts, nControls, controlPts, B, hnd
   end
   methods
      function obj = GpuPWLinearControl(t, nControlPts, nControls)           % PWLinearControl.m:13
         obj.hnd = libpointer('voidPtrPtr');
         ocs_check(calllib('libocs', 'ocs_control_create', obj.hnd, 1, t, numel(t), nControlPts, nControls));
         obj.nControlPts = nControlPts;  obj.nControls = nControls;
         obj.B = zeros(nControlPts, numel(t));  obj.controlPts = zeros(1, nControlPts);
         [~, ~, obj.B] = calllib('libocs', 'ocs_control_basis', obj.hnd.Value, obj.B);
         [~, ~, obj.controlPts] = calllib('libocs', 'ocs_control_points', obj.hnd.Value, obj.controlPts);
      end
      function u = compute_u(obj, v)                                          % :59
         batch = size(v, 2);  u = zeros(obj.nControls, size(obj.B, 2), batch);
         [~, ~, ~, u] = calllib('libocs', 'ocs_control_compute_u', obj.hnd.Value, batch, v, u);
      end
      function dJdv = compute_dJdv(obj, dJdu)                                 % :53
         batch = size(dJdu, 3);  dJdv = zeros(obj.nControls * obj.nControlPts, batch);
         [~, ~, ~, dJdv] = calllib('libocs', 'ocs_control_compute_dJdv', obj.hnd.Value, batch, dJdu, dJdv);
      end
      function v = compute_initial_v(obj, u0)                                 % :65
         v = zeros(obj.nControls * obj.nControlPts, 1);
         [rc, ~, ~, v] = calllib('libocs', 'ocs_control_compute_initial_v', obj.hnd.Value, u0, numel(u0), v);
         ocs_check(rc);
      end
      function [Lb, Ub] = compute_nlp_bounds(obj, controlBounds)              % :21
         Lb = zeros(obj.nControls * obj.nControlPts, 1);  Ub = Lb;
         [~, ~, ~, Lb, Ub] = calllib('libocs', 'ocs_control_compute_nlp_bounds', obj.hnd.Value, controlBounds, Lb, Ub);
      end
      function uFunc = compute_uFunc(obj, v)                                  % :74
         v = reshape(v, obj.nControls, []);
         uFunc = vectorInterpolant(obj.controlPts, v, 'linear');   % the reference's own wrapper
      end
   end
end
