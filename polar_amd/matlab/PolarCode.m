classdef PolarCode < handle
    % PolarCode — MATLAB host class over the MI355X polar decoder (C-ABI include/polar_amd.h via the
    % MEX gateway polar_mex.cpp). Same constructor and method names/argument order as the reference
    % class PolarM/PolarCode.m (ctor :59, encode :266, decode_sc_p1 :290, decode_scl_p1 :299,
    % decode_scl_llr :312, get_bler_quick :781), so existing BLER scripts (PolarM/main.m) run unchanged.
    % Row vectors of doubles in and out, like the reference.
    %
    % Build the gateway once:  mex -I<repo>/include polar_mex.cpp -L<repo>/polar_amd -lpolar_amd
    properties
        block_length
        info_length
        crc_size
        n
        design_epsilon
        frozen_bits      % 1 x N, in decoding order (as PolarM)
        info_bits        % 1-based positions of the K+crc unfrozen bits, most reliable first
        crc_matrix
    end
    properties (Access = private)
        h                % uint64 handle owned by the MEX gateway
    end
    methods
        function obj = PolarCode(block_length, info_length, design_epsilon, crc_size)
            if nargin < 4
                crc_size = 0;
            end
            obj.block_length = block_length;
            obj.info_length = info_length;
            obj.n = log2(block_length);
            obj.design_epsilon = design_epsilon;
            obj.crc_size = crc_size;
            obj.h = polar_mex('create', obj.n, info_length, design_epsilon, crc_size);
            [fz, order, crcm] = polar_mex('tables', obj.h);
            obj.frozen_bits = double(fz(:)');
            obj.info_bits = double(order(1:info_length + crc_size)) + 1;
            obj.crc_matrix = double(crcm);
        end
        function delete(obj)
            if ~isempty(obj.h)
                polar_mex('destroy', obj.h);
                obj.h = [];
            end
        end
        function coded_bits = encode(obj, info_bits)
            coded_bits = double(polar_mex('encode', obj.h, uint8(info_bits(:)')));
        end
        function decoded_bits = decode_sc_p1(obj, p1)
            decoded_bits = double(polar_mex('decode_sc_p1', obj.h, double(p1(:)')));
        end
        function u = decode_scl_p1(obj, p1, p0, list_size)
            u = double(polar_mex('decode_scl_p1', obj.h, double(p1(:)'), double(p0(:)'), list_size));
        end
        function u = decode_scl_llr(obj, llr, list_size)
            % llr may be 1 x N or B x N (one codeword per row): rows are decoded as one GPU batch
            u = double(polar_mex('decode_scl_llr', obj.h, double(llr), list_size));
        end
        function [bler, ber] = get_bler_quick(obj, ebno_vec, list_size_vec, max_runs, max_err, seed)
            % bler(i_ebno, i_list) as PolarM (:781-850); PolarM constants max_err=50, max_runs=500 (:788-789)
            if nargin < 4, max_runs = 500; end
            if nargin < 5, max_err = 50; end
            if nargin < 6, seed = 1; end
            b = polar_mex('get_bler_quick', obj.h, double(ebno_vec(:)'), uint8(list_size_vec(:)'), max_runs, max_err, seed);
            bler = b';      % gateway returns [n_L x n_e] (PolarC layout); PolarM indexes (ebno, list)
            ber = [];       % the reference's BER output is not produced by the GPU engine
        end
        % names used by the project brief
        function u = decode_SC_P1(obj, p1), u = obj.decode_sc_p1(p1); end
        function u = decode_SCL_P1(obj, p1, p0, L), u = obj.decode_scl_p1(p1, p0, L); end
        function u = decode_SCL_LLR(obj, llr, L), u = obj.decode_scl_llr(llr, L); end
    end
end
